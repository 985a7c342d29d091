"""Resampler front end: CLI exit codes (CPU) and the port-8572 HTTP protocol with batching (GPU)."""
import http.client
import threading
import wave

import numpy as np
import pytest

from goofer_amd import cli
from goofer_amd import sampler as S
from goofer_amd import synthetic as syn


def test_cli_argument_errors_exit_1(caplog):
    assert cli.main(["a.wav", "b.wav", "C4"]) == 1            # < 13 args -> TypeError -> help, exit 1
    assert "Expected 13 arguments but got 3" in caplog.text
    assert cli.main(["x_features.goofy"]) == 1                # editor mode: out of scope, loud
    # a bare flag letter crashes the flag arithmetic with TypeError exactly like the reference
    assert cli.main(["a.wav", "b.wav", "C4", "100", "g", "0", "1000", "0", "0", "100", "0", "!120", "AA"]) == 1


def test_split_arguments_keeps_spaces_in_paths():
    from conftest import golden
    g = golden("flags_pitch")
    assert S.split_arguments(str(g["split_in"][0])) == [str(v) for v in g["split_out"]]      # the reference's own answer
    with pytest.raises(ValueError):
        S.split_arguments("only_one.wav C4 100 g0 0 1000 0 700 100 0 !120 AA")


@pytest.mark.gpu
def test_http_server_batches_concurrent_requests(tmp_path):
    torch = pytest.importorskip("torch")
    from goofer_amd import core
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer
    ctx = Context(0)
    collector = cli.BatchCollector(Renderer(ctx), window_s=0.25)
    httpd, _ = cli.serve(0, collector, host="127.0.0.1")
    port = httpd.server_address[1]
    th = threading.Thread(target=httpd.serve_forever, daemon=True)
    th.start()
    try:
        bodies = []
        for i in range(6):
            src = syn.make_source(100 + i, seconds=0.4)
            wav = tmp_path / f"s{i}.wav"
            core.save_features(wav.with_name(f"s{i}_features.goofy"), src["env_pack"], src["f0"], src["mask"], src["formants"],
                               src["sr"], src["y_len"])
            req = syn.make_request(100 + i, ["t0g0", "g30", "L1", "fa20fb-10", "br30", "V80B10"][i], length_ms=200 + 40 * i)
            bodies.append(" ".join([str(wav), str(tmp_path / f"o{i}.wav")] + syn.request_args(req)))
        bodies.append(" ".join([str(tmp_path / "missing.wav"), str(tmp_path / "bad.wav")] + syn.request_args(req)))   # no features
        status = [None] * len(bodies)
        text = [None] * len(bodies)

        def post(k):
            c = http.client.HTTPConnection("127.0.0.1", port, timeout=120)
            c.request("POST", "/", body=bodies[k].encode("utf-8"))
            r = c.getresponse()
            status[k], text[k] = r.status, r.read().decode()
            c.close()

        ts = [threading.Thread(target=post, args=(k,)) for k in range(len(bodies))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(180)
        assert status[:6] == [200] * 6 and all(t == "" for t in text[:6])
        assert status[6] == 500 and text[6].startswith("An error occurred.\n") and "FileNotFoundError" in text[6]
        assert max(collector.batches) >= 2                    # concurrent requests shared a device batch
        # the audio itself: the same six requests rendered directly (another Philox seed: the noise stems differ sample by
        # sample, the harmonic stem — most of the energy — does not), compared through level and the low-band waveform
        from goofer_amd.render import Source
        direct = Renderer(ctx).render([(Source.from_pack(*core.load_features(str(tmp_path / f"s{i}_features.goofy"))),
                                        S.decode_request(*S.split_arguments(bodies[i])[2:])) for i in range(6)], seed=12345)
        for i in range(6):
            with wave.open(str(tmp_path / f"o{i}.wav"), "rb") as w:
                assert w.getframerate() == 44100 and w.getnframes() == int(0.1 * 44100) + int((0.2 + 0.04 * i) * 44100)
                assert w.getsampwidth() == 2
                raw = w.readframes(w.getnframes())
            got = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
            want = np.asarray(direct[i], dtype=np.float64)
            assert got.shape == want.shape and np.isfinite(got).all() and np.abs(got).max() > 1e-3
            r_got, r_want = np.sqrt(np.mean(got ** 2)), np.sqrt(np.mean(want ** 2))
            assert abs(r_got - r_want) <= 0.1 * r_want, (i, r_got, r_want)
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
        c.request("GET", "/")
        assert c.getresponse().status == 200
    finally:
        httpd.shutdown()
        collector.close()
        ctx.close()


def test_malformed_features_are_rejected_before_upload():
    """A short mask, a knot table that disagrees with its frequencies, a non-numeric formant track: Source.from_pack refuses
    them on the host (they would shift every later note's offsets in a device batch)."""
    import numpy as np
    import pytest
    from goofer_amd import synthetic as syn
    from goofer_amd.render import Source
    src = syn.make_source(5, seconds=0.2)
    ok = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    assert ok.knots.shape[1] == 1 + src["y_len"] // 256
    with pytest.raises(ValueError, match="voicing mask"):
        Source.from_pack(src["env_pack"], src["f0"], src["mask"][:-10], src["formants"], src["sr"], src["y_len"])
    bad = dict(src["env_pack"])
    bad["hz_knots"] = bad["hz_knots"][:-1]
    with pytest.raises(ValueError, match="knot table"):
        Source.from_pack(bad, src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    with pytest.raises(ValueError, match="formant"):
        Source.from_pack(src["env_pack"], src["f0"], src["mask"], {1: ["a", "b"]}, src["sr"], src["y_len"])
    with pytest.raises(ValueError):
        Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], 0)
    no_f1 = {k: v for k, v in src["formants"].items() if k != 1}          # F2..F4 would silently take F1..F3's places
    with pytest.raises(KeyError):
        Source.from_pack(src["env_pack"], src["f0"], src["mask"], no_f1, src["sr"], src["y_len"])


def test_server_binds_to_loopback_by_default_and_worker_survives_errors():
    from goofer_amd import cli

    class Boom:
        def render(self, jobs, seed=0):
            raise MemoryError("device gone")

    col = cli.BatchCollector(renderer=Boom(), window_s=0.0)
    httpd, _ = cli.serve(port=0, collector=col)
    try:
        assert httpd.server_address[0] == "127.0.0.1"
        import pytest
        with pytest.raises(Exception):
            col.submit(["a.wav", "b.wav"] + ["0"] * 11)
        with pytest.raises(Exception):                       # the worker thread is still there for the next request
            col.submit(["a.wav", "b.wav"] + ["0"] * 11)
        assert col._thread.is_alive()
    finally:
        httpd.server_close()
        col.close()


def test_server_keeps_loaded_features_until_the_file_changes(tmp_path):
    """The collector's feature cache: the same voicebank sample asked for again is the same Source object (so a batch uploads
    it once); a rewritten file is loaded again; the cache is bounded."""
    import os
    from goofer_amd import core
    col = cli.BatchCollector(renderer=object(), max_sources=2)
    try:
        paths = []
        for i in range(3):
            src = syn.make_source(300 + i, seconds=0.3)
            f = tmp_path / f"v{i}_features.goofy"
            core.save_features(f, src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
            paths.append(f)
        a = col._source(paths[0])
        assert col._source(paths[0]) is a and a.ylen == int(round(0.3 * 44100))
        src = syn.make_source(999, seconds=0.4)
        core.save_features(paths[0], src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
        os.utime(paths[0], ns=(1, 1))                       # whatever the clock granularity: the stamp differs
        b = col._source(paths[0])
        assert b is not a and b.ylen == int(round(0.4 * 44100))
        col._source(paths[1]); col._source(paths[2])
        assert len(col._sources) == 2 and str(paths[0]) not in col._sources
    finally:
        col.close()
