"""Resampler front end: the reference's CLI and port-8572 HTTP protocol (SillySampler.py:1187-1275).

  python -m goofer_amd.cli in.wav out.wav pitch velocity flags offset length consonant cutoff volume
                           modulation !tempo pitch_string          -> render one note, exit 0 / 1
  python -m goofer_amd.cli                                          -> HTTP server on port 8572

HTTP: ``POST /`` with the 13 arguments joined by single spaces (wav paths may contain spaces) answers
``200`` (empty) when ``out.wav`` is written or ``500 text/plain`` ``An error occurred.\\n<traceback>``;
``GET`` answers ``200`` (liveness probe).  Unlike the reference, which renders each request inline in its
thread, requests that arrive within a few milliseconds are rendered as ONE GPU batch (``BatchCollector``);
a request that fails (bad flags, missing features, ...) is isolated and does not fail its batch-mates.
"""
from __future__ import annotations

import collections
import logging
import sys
import threading
import traceback
from http.server import BaseHTTPRequestHandler, HTTPServer
from pathlib import Path
from socketserver import ThreadingMixIn

import numpy as np

from . import sampler as S
from . import trackers

HELP = (
    "Usage:\n"
    "  SillySampler.py in.wav out.wav pitch velocity flags\n"
    "           offset(ms) length(ms) consonant(ms) cutoff(ms)\n"
    "           volume(%) modulation(%) !tempo pitch_string\n\n"
    "Example:\n"
    "  SillySampler.py in.wav out.wav C4 100 g0 0 1000 0 700 100 0 !120 AA"
)


class _Pending:
    __slots__ = ("args", "done", "error")

    def __init__(self, args):
        self.args, self.done, self.error = args, threading.Event(), None


class BatchCollector:
    """Gathers concurrent render requests and runs them as device batches — up to ``lanes`` batches in flight: each lane is a
    Renderer with its own library handle and HIP stream (the voicebank arena is shared), so while one batch is on the device
    or being written out, the requests that arrived meanwhile are decoded, planned and rendered on the other lane
    (SillySampler.py:1196-1224 answers every request with its own render; a song is thousands of them)."""

    def __init__(self, renderer=None, window_s: float = 0.005, max_batch: int = 4096, max_sources: int = 512, tracker=None,
                 lanes: int = 2):
        self._renderer = renderer                           # an injected renderer is the only lane
        self._n_lanes = 1 if renderer is not None else max(1, int(lanes))
        self._lanes = None                                  # made on first use: [(Renderer, stream or None)]
        self._free = None
        self.tracker = tracker                              # f0 / formant tracker of the cold-cache path (goofer_amd.trackers.get)
        self.window_s, self.max_batch, self.max_sources = window_s, max_batch, max_sources
        self._sources = collections.OrderedDict()          # feature cache: path -> ((mtime_ns, size), Source)
        self._src_lock = threading.Lock()
        self._fresh = 0                                        # samples cached since the collector's old generation was last frozen (see _source)
        self._lock = threading.Condition()
        self._queue = []
        self._stop = False
        self._thread = threading.Thread(target=self._loop, name="goofer-batch", daemon=True)
        self._thread.start()
        self.batches = []           # sizes of the batches rendered (observability / tests)

    @property
    def renderer(self):
        return self._lane_list()[0][0]

    def _lane_list(self):
        if self._lanes is None:
            import queue
            if self._renderer is not None:
                self._lanes = [(self._renderer, None)]
            else:
                import torch
                from .device import Context
                from .render import Renderer
                first = Renderer()
                self._lanes = [(first, torch.cuda.Stream(first.ctx.device))]
                for _ in range(self._n_lanes - 1):
                    r = Renderer(Context(first.ctx.device.index if first.ctx.device.index is not None else 0), hop=first.hop)
                    r.sources = first.sources
                    self._lanes.append((r, torch.cuda.Stream(r.ctx.device)))
            self._free = queue.Queue()
            for ln in self._lanes:
                self._free.put(ln)
        return self._lanes

    def submit(self, args) -> None:
        """Blocks until the note is written; raises what the render raised."""
        p = _Pending(list(args))
        with self._lock:
            self._queue.append(p)
            self._lock.notify_all()
        while not p.done.wait(1.0):
            if not self._thread.is_alive():
                raise RuntimeError("the render worker has stopped")
        if p.error is not None:
            raise p.error

    def close(self):
        with self._lock:
            self._stop = True
            self._lock.notify_all()
        self._thread.join(5)
        if self._lanes is not None:                         # batches still on a lane finish (their waiters get their answers)
            import queue
            held = []
            try:
                for _ in self._lanes:
                    held.append(self._free.get(timeout=30))
            except queue.Empty:
                pass
            for ln in held:
                self._free.put(ln)

    def _loop(self):
        while True:
            with self._lock:
                while not self._queue and not self._stop:
                    self._lock.wait()
                if self._stop and not self._queue:
                    return
                self._lock.wait(self.window_s)            # let a burst accumulate
                batch, self._queue = self._queue[:self.max_batch], self._queue[self.max_batch:]
            try:
                self._lane_list()
                lane = self._free.get()                    # waits while every lane holds a batch
            except BaseException as e:   # noqa: BLE001
                self._fail(batch, e)
                continue
            if len(self._lanes) == 1:
                self._run_on(lane, batch)
            else:
                threading.Thread(target=self._run_on, args=(lane, batch), name="goofer-lane", daemon=True).start()

    def _fail(self, batch, e):
        for p in batch:
            if not p.done.is_set():
                p.error = e if isinstance(e, Exception) else RuntimeError(f"render worker: {e!r}")
                p.done.set()

    def _run_on(self, lane, batch):
        try:
            if lane[1] is not None:
                import torch
                with torch.cuda.stream(lane[1]):
                    self._render(batch, lane[0])
            else:
                self._render(batch, lane[0])
        except BaseException as e:       # noqa: BLE001 - nothing may kill the worker: every waiter gets an answer
            self._fail(batch, e)
        finally:
            self._free.put(lane)

    def _source(self, feat: Path):
        """Features of one voicebank sample, kept across requests (a song asks for the same samples over and over): keyed by
        path, dropped when the file changes (size / mtime), at most ``max_sources`` of them.  The same Source object in several
        notes of a batch is also what lets the renderer upload it once."""
        from . import core
        from .render import Source
        st = feat.stat()
        key, stamp = str(feat), (st.st_mtime_ns, st.st_size)
        with self._src_lock:
            hit = self._sources.get(key)
            if hit is not None and hit[0] == stamp:
                self._sources.move_to_end(key)
                return hit[1]
        env, f0, mask, forms, sr, ylen = core.load_features(feat)
        src = Source.from_pack(env, f0, mask, forms, sr, ylen)
        with self._src_lock:
            self._sources[key] = (stamp, src)
            while len(self._sources) > self.max_sources:
                self._sources.popitem(last=False)
            self._fresh += 1
            if self._fresh >= 256:                             # the cache holds them until their files change: out of the collector's way
                self._fresh = 0                                # (see main(): a full collection over a resident voicebank stalls a batch)
                import gc
                gc.freeze()
        return src

    def _render(self, batch, renderer=None):
        from .render import write_wav
        renderer = renderer or self.renderer
        jobs, owners = [], []
        for p in batch:                                    # per-note decode / feature load: errors stay per note
            try:
                if len(p.args) < 13:
                    raise TypeError(f"Expected 13 arguments but got {len(p.args)}")
                in_file, out_file = Path(p.args[0]), Path(p.args[1])
                req = S.decode_request(*p.args[2:13])
                # cached features, or analysed from the wav and cached on the first request for a sample (SillySampler.py:415-432)
                feat = trackers.features_path(in_file)
                if not feat.exists():
                    feat = trackers.ensure_features(in_file, hop_length=renderer.hop, tracker=self.tracker, ctx=renderer.ctx)
                src = self._source(feat)
                jobs.append((src, req))
                owners.append((p, out_file, src.sr))
            except Exception as e:      # noqa: BLE001 - reported to the client as a 500
                p.error = e
                p.done.set()
        groups = {}
        for j, (job, own) in enumerate(zip(jobs, owners)):
            groups.setdefault((job[0].sr, job[0].n_fft), []).append(j)
        for idxs in groups.values():
            try:
                outs = renderer.render([jobs[j] for j in idxs], seed=int(np.random.SeedSequence().generate_state(1)[0]))
                results = dict(zip(idxs, outs))
                errors = {}
            except Exception:           # noqa: BLE001 - isolate the offender by rendering one by one
                results, errors = {}, {}
                for j in idxs:
                    try:
                        results[j] = renderer.render([jobs[j]], seed=int(np.random.SeedSequence().generate_state(1)[0]))[0]
                    except Exception as e:   # noqa: BLE001
                        errors[j] = e
            self.batches.append(len(idxs))
            for j in idxs:
                p, out_file, sr = owners[j]
                try:
                    if j in errors:
                        raise errors[j]
                    logging.info(f"Writing {out_file}")
                    write_wav(out_file, results[j], sr)
                except Exception as e:  # noqa: BLE001
                    p.error = e
                p.done.set()


class ThreadedHTTPServer(ThreadingMixIn, HTTPServer):
    daemon_threads = True


def make_handler(collector: BatchCollector):
    class RequestHandler(BaseHTTPRequestHandler):
        def log_message(self, fmt, *a):                    # keep stdout quiet like the reference's logging level
            logging.debug(fmt % a)

        def do_GET(self):
            self.send_response(200)
            self.end_headers()

        def do_POST(self):
            body = self.rfile.read(int(self.headers["Content-Length"])).decode("utf-8")
            try:
                collector.submit(S.split_arguments(body))
            except Exception:           # noqa: BLE001
                self.send_response(500)
                self.send_header("Content-type", "text/plain")
                self.end_headers()
                self.wfile.write(f"An error occurred.\n{traceback.format_exc()}".encode("utf-8"))
                return
            self.send_response(200)
            self.end_headers()

    return RequestHandler


def serve(port: int = 8572, collector: BatchCollector | None = None, host: str = "127.0.0.1"):
    """The resampler server.  The reference listens on every interface (SillySampler.py:1220-1224); a request names the file it
    reads (a pickle-bearing .goofy) and the file it overwrites, so this one binds to the loopback address OpenUtau connects
    to and leaves other interfaces to an explicit ``--host``."""
    collector = collector or BatchCollector()
    httpd = ThreadedHTTPServer((host, port), make_handler(collector))
    return httpd, collector


def main(argv=None) -> int:
    logging.basicConfig(format="%(message)s", level=logging.INFO)
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.info(f"SillySampler {S.VERSION} (goofer_amd / MI355X)")
    if not argv or argv[0] == "--host":
        host = argv[1] if len(argv) > 1 else "127.0.0.1"
        httpd, _ = serve(host=host)
        # A server keeps a voicebank's samples and its own set-up alive for hours: a full collection that walks them is 50-100 ms
        # in the middle of a batch whose device work is 2 ms.  What exists now moves to the permanent generation; the young
        # generations still collect the per-request garbage.
        import gc
        gc.collect()
        gc.freeze()
        print(f"Starting HTTP server on port 8572 ({host or 'all interfaces'})...")
        httpd.serve_forever()
        return 0
    logging.info(f"Args: {argv} (count={len(argv)})")
    try:
        if all(Path(a).suffix.lower() == ".goofy" for a in argv):
            raise NotImplementedError("the Tk voicing editor is out of scope for the GPU backend")
        if len(argv) == 1 and Path(argv[0]).exists():                # folder (or single file) feature extraction   :1252-1263
            from .render import Renderer
            tally = trackers.extract_folder(argv[0], ctx=Renderer().ctx)
            return 0 if tally["failed"] == 0 else 1
        if len(argv) < 13:
            raise TypeError(f"Expected 13 arguments but got {len(argv)}")
        from .render import GooferResampler
        GooferResampler(*argv[:13])
    except TypeError as e:
        logging.error("Argument parsing failed: %s", str(e))
        logging.error(HELP)
        return 1
    except Exception:                   # noqa: BLE001
        logging.exception("Failed to render")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
