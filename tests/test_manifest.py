"""f4: the OpenUtau manifest is generated from the product's flag table (goofer_amd/flags.py); the committed SillySampler.yaml
is that output; every flag of the table is one the decoder understands with the table's default; and — in the build container,
where the reference is mounted — the manifest equals the reference's SillySampler.yaml as parsed data."""
import os

import pytest
import yaml

from goofer_amd import flags as F
from goofer_amd import sampler as S

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/SillySampler.yaml"


def test_committed_manifest_is_the_generated_one():
    text = open(os.path.join(REPO, "SillySampler.yaml")).read()
    assert text == F.manifest_yaml()
    assert yaml.safe_load(text) == F.manifest()
    assert len(F.manifest()["expressions"]) == 31


def test_launchers_point_at_the_entry_script():
    sh = open(os.path.join(REPO, "SillySampler.sh")).read()
    bat = open(os.path.join(REPO, "SillySampler.bat")).read()
    assert sh.startswith("#!/bin/sh") and 'SillySampler.py" "$@"' in sh
    assert "python SillySampler.py %*" in bat
    assert os.access(os.path.join(REPO, "SillySampler.sh"), os.X_OK)


def test_every_table_flag_decodes_with_its_range_and_default():
    base = ("C4", "100", "", "50", "1000", "100", "-250", "80", "0", "!125", "AA")
    r0 = S.decode_request(*base)
    for f in F.FLAGS:
        for v in (f.lo, f.hi, f.default):
            args = list(base)
            args[2] = f"{f.flag}{v}"
            r = S.decode_request(*args)                        # decodes without error over the whole advertised range
            got = S.parse_flags(args[2])
            assert got == {f.flag: v}
        # the table's default is the decoder's behaviour when the flag is absent
        args = list(base)
        args[2] = f"{f.flag}{f.default}"
        rd = S.decode_request(*args)
        field = f.drives.split("[")[0]
        if field != "flags":
            assert getattr(rd, field) == getattr(r0, field), f.flag
    # option expressions spell flag + digit
    for f in F.FLAGS:
        for o in f.options:
            assert o.startswith(f.flag) and o[len(f.flag):].isdigit()


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference is only mounted in the build container")
def test_manifest_equals_the_reference_manifest_as_data():
    ref = yaml.safe_load(open(REF))
    assert F.manifest() == ref
    assert list(F.manifest()["expressions"]) == list(ref["expressions"])      # same order, as OpenUtau lists them
