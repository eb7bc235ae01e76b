"""The tuning table (include/ccvpe.h ccvpe_import_tuning / ccvpe_export_tuning, ccvpe_amd/tuning.py): text format and file
merge on the CPU; on the GPU, two fresh processes that share a table return bit-identical outputs and the second one does
not measure anything."""
import json
import os
import subprocess
import sys

import pytest

from ccvpe_amd import tuning

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LINES = ["op v3_r20_c0_p0_b1|loc6.conv_a|256x640x12096|1100 conv_wino4_16x128 3",
         "op v3_r20_c0_p0_b1|grd.head|40x1280x320|0010 conv_igemm_64x64_m16 0"]


def test_parse_render_round_trip_and_foreign_lines_are_dropped():
    text = "# comment\n" + LINES[0] + "\nplan legacy\nop broken line\n" + LINES[1] + "\n"
    tab = tuning.parse(text)
    assert sorted(tab) == sorted(l.split()[1] for l in LINES)
    assert tuning.parse(tuning.render(tab)) == tab
    assert tuning.render({}) == ""


def test_save_merges_with_the_file_and_replaces_same_key_entries(tmp_path, monkeypatch):
    path = str(tmp_path / "sub" / "tuning.txt")
    monkeypatch.setattr(tuning, "export", lambda lib, h: LINES[0] + "\n")
    assert tuning.save_from(None, None, path) == path
    newer = LINES[0].rsplit(" ", 2)[0] + " conv_wino4_16x64 255"
    monkeypatch.setattr(tuning, "export", lambda lib, h: newer + "\n" + LINES[1] + "\n")
    tuning.save_from(None, None, path)
    tab = tuning.parse(open(path).read())
    assert len(tab) == 2 and tab[LINES[0].split()[1]] == newer
    assert not [f for f in os.listdir(tmp_path / "sub") if ".tmp." in f]


def test_user_cache_switch(monkeypatch):
    for off in ("", "0", "off"):
        monkeypatch.setenv("CCVPE_TUNE_CACHE", off)
        assert tuning.user_cache_path() is None
    monkeypatch.setenv("CCVPE_TUNE_CACHE", "/tmp/x.txt")
    assert tuning.user_cache_path() == "/tmp/x.txt"


def test_committed_table_parses_if_present():
    if os.path.exists(tuning.COMMITTED):
        text = open(tuning.COMMITTED).read()
        tab = tuning.parse(text)
        assert tab and all(len(v.split()) == 4 for v in tab.values())


CHILD = r"""
import json, sys, time, hashlib
import torch
sys.path.insert(0, %(root)r)
from ccvpe_amd import models, weights, _lib
sd = weights.generate_state_dict("oxford", 4)
g, s = weights.generate_inputs("oxford", 2, 4)
m = models.CVM_OxfordRobotCar("cuda"); m.load_state_dict(sd); m.to("cuda").eval()
g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter(); outs = m(g, s); torch.cuda.synchronize(); first = time.perf_counter() - t0
gen = _lib.load().ccvpe_tuning_generation(m._handle)
dig = [hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest() for o in outs]
print(json.dumps({"first_forward_s": first, "tuned_plans": gen, "digests": dig, "entries": len(m.export_tuning().splitlines())}))
"""


@pytest.mark.gpu
def test_two_fresh_processes_with_one_table_are_bit_identical(tmp_path):
    env = dict(os.environ, CCVPE_TUNE_CACHE=str(tmp_path / "tuning.txt"), CCVPE_TUNE_IGNORE_COMMITTED="1")
    recs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        recs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    a, b = recs
    assert a["tuned_plans"] >= 1 and b["tuned_plans"] == 0, "the second process finds every launch in the table"
    assert a["digests"] == b["digests"], "same table -> same launches -> same bits"
    assert b["entries"] >= a["entries"] > 0
    assert b["first_forward_s"] < a["first_forward_s"], (a, b)
