"""The walk farm's round protocol (csrc/farm64.hip.h: farm_predict + the two sets of the overlapped rounds) as a CPU MODEL, tools/farm_protocol_sim.cpp:
a noisy diagonal path through a 1 Mb x 1 Mb pair, walked in rounds of tiles asked for one round (plain) or two rounds (overlapped) ahead.
Guards the property the device code relies on -- after a wrong guess the two sets are in step again within a round -- against the first protocol
(skip the other set's tiles wherever they are), which fell to ~3 tiles a round.  The device kernels themselves are checked against the oracle in
tests/test_long_range.py (-m gpu); this model shares their prediction code by restatement only."""
import os
import re
import subprocess

import common


def _rounds(exe, *args):
    out = subprocess.check_output([exe] + [str(a) for a in args]).decode()
    m = re.search(r"rounds (\d+) tiles (\d+)", out)
    return int(m.group(1)), int(m.group(2))


def test_overlapped_rounds_stay_in_step(tmp_path):
    src = os.path.join(common.HERE, "..", "tools", "farm_protocol_sim.cpp")
    exe = str(tmp_path / "farm_sim")
    subprocess.check_call(["g++", "-O2", "-o", exe, src])
    for nt in (8, 16, 32):
        plain, tiles = _rounds(exe, nt, 0)
        over, tiles2 = _rounds(exe, nt, 1)
        assert tiles == tiles2
        assert plain <= 1.05 * tiles / nt + 2          # nearly every guess along the line is right
        assert over <= 1.03 * plain + 2                 # a wrong guess costs the rest of one round, not the protocol
    first, _ = _rounds(exe, 16, 1, 1)                   # the first protocol (kept in the model): sets become each other's complements
    assert first > 2 * _rounds(exe, 16, 1)[0]


def test_long_pair_roofline_bytes():
    """bench.py's long_pairs leg: the algorithmic bytes of a one-pair sweep (1 Mb x 1 Mb = 12.5 GB of bottom rows written + read, 18.8 GB of snapshots)"""
    import sys
    sys.path.insert(0, os.path.join(common.HERE, ".."))
    import bench
    r = bench.long_pair_roofline({"fn": "AffineGap(HumanChimpTwo,-600,-150)", "n": 1000000, "m": 999886, "cells": 999886000000, "sweep_ms": 312.0})
    assert r["waves"] == 1563 and abs(r["algorithmic_bytes_per_launch"] - 43.74e9) < 0.05e9
    assert abs(r["frac"] - 43.74e9 / 0.312 / 8e12) < 1e-4
    c = bench.long_pair_roofline({"fn": "ConstGap(HumanChimpTwo,-430)", "n": 150000, "m": 180009, "cells": 27001350000, "sweep_ms": 25.7})
    assert c["waves"] == 235 and c["algorithmic_bytes_per_launch"] == 4 * 180010 * 234 * 2 + (180072 // 224) * 235 * 64 * 12 * 4
