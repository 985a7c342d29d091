"""Render a handful of notes as ONE GPU batch through the reference's call surface (needs an MI355X).

    python examples/render_batch.py out_dir

Each note is what UTAU / OpenUtau would send to SillySampler.py: a source sample (its ``<stem>_features.goofy`` cache, here
synthetic) plus the 11 resampler arguments.  ``Renderer.render`` plans the notes on the host and runs
goofer_assemble_batch -> goofer_synth_batch (-> goofer_post_batch when a note carries post-chain flags) once for all."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from goofer_amd import sampler as S
from goofer_amd import synthetic as syn
from goofer_amd.render import Renderer, Source, write_wav


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    renderer = Renderer()
    requests = [
        ("C4", "100", "g-10", "30", "600", "80", "50", "100", "0", "!120", "AA"),               # plain, slightly darker
        ("E4", "100", "t30fa20fb-10B30", "30", "900", "80", "50", "90", "0", "!120", "AA#20#AP"),   # formant edits, pitch bend
        ("G4", "80", "L1sg40st-30", "30", "1200", "80", "50", "100", "0", "!120", "AA"),         # mirrored loop, sub-harmonics, soft
        ("A3", "120", "vf30sa20pd40", "30", "700", "80", "50", "100", "0", "!120", "AA#10#BA#10#AA"),   # fry, whisper blend, pitch dynamics
    ]
    jobs = []
    for i, args in enumerate(requests):
        src = syn.make_source(100 + i, seconds=0.5)                    # stands in for core.load_features(<stem>_features.goofy)
        jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*args)))
    audio = renderer.render(jobs, seed=1)
    for i, (y, (source, _)) in enumerate(zip(audio, jobs)):
        path = os.path.join(out_dir, f"note{i}.wav")
        write_wav(path, y, source.sr)
        print(f"{path}: {len(y)} samples, peak {abs(y).max():.3f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "rendered")
