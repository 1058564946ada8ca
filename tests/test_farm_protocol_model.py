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
    """bench.py's long_pairs leg: `frac` prices the sweep with SURVEY 8d's bytes like every other leg -- n + m + ceil(b n m / 8) + ceil(b (n + m) / 8) + 8 + 16 |cigar| --
    and `traffic_model` holds the bytes this design moves (bottom rows written + read, snapshots), for the rows per lane / snapshot spacing the call ran with"""
    import sys
    sys.path.insert(0, os.path.join(common.HERE, ".."))
    import bench
    d = {"fn": "AffineGap(HumanChimpTwo,-600,-150)", "n": 1000000, "m": 999886, "cells": 999886000000, "sweep_ms": 312.0, "call_s": 0.35, "runs": 3839}
    r = bench.long_pair_roofline(d)  # (no geometry given: round 5's 10 rows per lane, a snapshot every 512 steps)
    alg = 1000000 + 999886 + (6 * 1000000 * 999886 + 7) // 8 + (6 * 1999886 + 7) // 8 + 8 + 16 * 3839
    assert r["algorithmic_bytes_per_launch"] == alg and abs(r["frac"] - alg / 0.312 / 8e12) < 1e-6 and abs(r["frac"] - 0.30) < 0.005
    assert abs(r["frac_call"] - alg / 0.35 / 8e12) < 1e-6
    assert r["waves"] == 1563 and abs(r["traffic_model"]["bytes_per_launch"] - 43.74e9) < 0.05e9
    r8 = bench.long_pair_roofline(dict(d, sweep_ms=259.5), (8, 512))  # round 6: 8 rows per lane for this pair
    assert r8["waves"] == 1954 and r8["traffic_model"]["rows_per_lane"] == 8
    assert r8["traffic_model"]["bytes_per_launch"] == 8 * 999887 * 1953 * 2 + (999949 // 512) * 1954 * 64 * 20 * 4
    c = bench.long_pair_roofline({"fn": "ConstGap(HumanChimpTwo,-430)", "n": 150000, "m": 180009, "cells": 27001350000, "sweep_ms": 25.7, "call_s": 0.036, "runs": 16378})
    assert c["waves"] == 235 and c["traffic_model"]["bytes_per_launch"] == 4 * 180010 * 234 * 2 + (180072 // 224) * 235 * 64 * 12 * 4
    assert c["algorithmic_bytes_per_launch"] == 150000 + 180009 + (2 * 150000 * 180009 + 7) // 8 + (2 * 330009 + 7) // 8 + 8 + 16 * 16378
    c4 = bench.long_pair_roofline({"fn": "ConstGap(HumanChimpTwo,-430)", "n": 150000, "m": 180009, "cells": 27001350000, "sweep_ms": 19.3, "call_s": 0.03, "runs": 16378}, (4, 224))
    assert c4["waves"] == 586 and c4["traffic_model"]["bytes_per_launch"] == 4 * 180010 * 585 * 2 + (180072 // 224) * 586 * 64 * 8 * 4
