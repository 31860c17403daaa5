"""CPU tier: the committed evidence files say what the documents say they say.

VERDICT r5: profiles/r05_parity_margins.json -- cited by README / BASELINE / DESIGN as the GPU tier's measured margins -- held the EMULATOR
tier's records at the end of round 5 (both tiers wrote gpurun_out/parity_margins.json and the CPU run came last).  The tiers write separate
files now (tests/helpers.py); this test fails when a committed profiles/rNN_parity_margins.json contains a record that was not measured on a GPU."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_parity_margins_are_gpu_records():
    files = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_parity_margins.json"))
             if re.fullmatch(r"r\d+_parity_margins\.json", os.path.basename(p))]
    assert files, "no committed parity-margin files"
    for p in files:
        recs = json.load(open(p))
        assert recs, p
        bad = [r.get("case") for r in recs if not str(r.get("device", "")).startswith("cuda")]
        assert not bad, "%s holds %d record(s) not measured on a GPU (e.g. %s): it is cited as the GPU tier's evidence" % (
            os.path.basename(p), len(bad), bad[:3])


def test_margin_files_are_separate_per_tier(tmp_path, monkeypatch):
    import helpers
    monkeypatch.delenv("DN_PARITY_MARGINS", raising=False)
    assert helpers.margins_path("cuda:0") != helpers.margins_path("cpu")
    assert helpers.margins_path("cuda:0").endswith("parity_margins_cuda.json")
    assert helpers.margins_path("cpu").endswith("parity_margins_emu.json")


def test_baseline_md_has_one_table_per_round():
    heads = [l for l in open(os.path.join(ROOT, "BASELINE.md")) if l.startswith("## ")]
    nums = [h.split(".")[0] for h in heads]
    assert len(nums) == len(set(nums)), "duplicate section numbers in BASELINE.md: %s" % nums
