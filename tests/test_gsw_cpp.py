"""The C++ mirror of the graph aligner's read path (include/gonomics_genomegraph.hpp: index, seeds, traversals as stack machines, the
per-read driver, rounds of batched device DPs through the raw C ABI) against the Python mirror and the literal restatement
tests/pyref_gsw.py.  Rows N2 (callers) / N4 of SURVEY 8f; parity unpinned by the reference (its tests only log)."""
import os
import subprocess

import numpy as np
import pytest

import common
import pyref_gsw as ref
from test_gsw_reads import build, make_case
from gonomics_amd import genomeGraph as gg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "gsw_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "gsw_mirror_test.bin")
LIB = os.path.join(ROOT, "gonomics_amd", "libgonomics_align_hip.so")
MX = common.matrices()


def _build():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
    # (the CPU backend + the oracle are linked for the "cpu" baseline mode of this TEST binary only)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-DGNX_TEST_BACKEND", "-I" + os.path.join(ROOT, "include"), "-o", BIN, SRC,
                           os.path.join(ROOT, "tests", "cpp", "gsw_cpu_backend.cpp"), LIB, os.path.join(ROOT, "oracle", "liboracle.so"),
                           "-Wl,-rpath," + os.path.join(ROOT, "gonomics_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-L/opt/rocm/lib", "-lamdhip64"])


def write_case(path, seqs, edges, reads, seed_len, seed_step, sc):
    with open(path, "w") as fh:
        fh.write("%d\n" % len(seqs))
        for s in seqs:
            fh.write("%d %s\n" % (len(s), " ".join(str(int(x)) for x in s)))
        fh.write("%d\n" % len(edges))
        for u, v in edges:
            fh.write("%d %d\n" % (u, v))
        fh.write("%d\n" % len(reads))
        for r in reads:
            fh.write("%d %s\n" % (len(r), " ".join(str(int(x)) for x in r)))
        fh.write("%d %d\n" % (seed_len, seed_step))
        fh.write(" ".join(str(int(x)) for x in np.asarray(sc, dtype=np.int64).reshape(25)) + "\n")


def read_out(path):
    rows, timing = [], None
    for line in open(path):
        if line.startswith("#"):
            timing = [float(x) for x in line[1:].split()]
            continue
        if line.strip() == "panic":
            rows.append("panic")
            continue
        parts = [x.strip() for x in line.split("|")]
        head, nodes, cig, seqlen = parts[:4]
        h = [int(x) for x in head.split()]
        c = None if cig == "none" else tuple((int(a), int(b)) for a, b in zip(cig.split()[0::2], cig.split()[1::2]))
        rows.append((h[0], h[1], bool(h[2]), h[3], tuple(int(x) for x in nodes.split()), h[4], c, h[5], int(seqlen)) + ((int(parts[4]),) if len(parts) > 4 else ()))
    return rows, timing


def test_cpp_gsw_mirror_builds_and_refuses_without_gpu(tmp_path):
    _build()
    seqs, edges, reads = make_case(7, "linear")
    write_case(str(tmp_path / "case.txt"), seqs, edges, reads[:2], 16, 1, MX["HumanChimpTwo"])
    rc = subprocess.call([BIN, str(tmp_path / "case.txt"), str(tmp_path / "out.txt")])
    assert rc in (0, 2)  # 2 == "no HIP device" (no CPU fallback); 0 on a GPU box


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed_len,step", [("linear", 16, 1), ("snp", 16, 1), ("snp", 20, 7), ("wide", 16, 1), ("wide3", 16, 1)])
def test_cpp_gsw_mirror_equals_python_mirror(gpu_lib, tmp_path, kind, seed_len, step):
    _build()
    seqs, edges, reads = make_case(9, kind)
    sc = MX["HumanChimpTwo"]
    write_case(str(tmp_path / "case.txt"), seqs, edges, reads, seed_len, step, sc)
    assert subprocess.call([BIN, str(tmp_path / "case.txt"), str(tmp_path / "out.txt")]) == 0
    rows, timing = read_out(str(tmp_path / "out.txt"))
    assert len(rows) == len(reads) and timing is not None
    # the benchmark's CPU-baseline mode (extension DPs on the CPU oracle, all host threads) walks the same read path to the same results
    assert subprocess.call([BIN, str(tmp_path / "case.txt"), str(tmp_path / "out_cpu.txt"), "reads", "cpu"]) == 0
    assert read_out(str(tmp_path / "out_cpu.txt"))[0] == rows
    g = build(seqs, edges)
    index = gg.SeedIndex(g.Nodes, seed_len, step)
    bigs = [gg.FastqBig("r%d" % k, rd) for k, rd in enumerate(reads)]
    py = gg.GswBatchToGiraf(g, bigs, index, seed_len, sc, on_panic="mark")
    nodes = ref.make_graph(seqs, edges)
    full = ref.index_genome(nodes, seed_len, step)
    panics = 0
    for k, r in enumerate(py):
        r2 = ref.make_read(reads[k])  # ... and the sequential restatement on the CPU oracle
        try:
            exp = ref.giraf_key(ref.read_to_giraf(nodes, r2, ref.seed_map(full, nodes, r2, seed_len), sc))
        except IndexError:  # the Go code panics on this read (search.go:139 with a short Prev node): all three must say so
            assert isinstance(r, gg.GoPanic) and rows[k] == "panic", "read %d" % k
            panics += 1
            continue
        key = r.key()
        assert rows[k] == key[:8] + (len(key[8]),), "read %d" % k
        assert key == exp, "read %d" % k
    assert (panics > 0) == (kind == "snp")  # bubbles of 1 .. 5 bases are what the reference cannot extend across; long alleles are fine


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["snp", "wide3"])
def test_cpp_gsw_worker_pool_equals_one_thread(gpu_lib, tmp_path, kind):
    """the mirror's worker pool (per-read host work on GNX_GSW_THREADS threads, genomeGraph/routines.go:12-65) returns what one thread
    returns, read for read, panics included"""
    _build()
    seqs, edges, reads = [], [], []
    for seed in (21, 22, 23, 24, 25):  # five cases side by side in one graph: 200 reads
        s2, e2, r2 = make_case(seed, kind)
        edges += [(u + len(seqs), v + len(seqs)) for u, v in e2]
        seqs += s2
        reads += r2
    write_case(str(tmp_path / "case.txt"), seqs, edges, reads, 16, 1, MX["HumanChimpTwo"])
    outs = {}
    for threads in ("1", "3", "16"):
        env = dict(os.environ, GNX_GSW_THREADS=threads)
        assert subprocess.call([BIN, str(tmp_path / "case.txt"), str(tmp_path / ("out%s.txt" % threads))], env=env) == 0
        outs[threads], timing = read_out(str(tmp_path / ("out%s.txt" % threads)))
        assert int(timing[3]) == int(threads)
    assert len(outs["1"]) == len(reads) and outs["1"] == outs["3"] == outs["16"]
    assert any(r != "panic" and r[7] > 0 for r in outs["1"])


@pytest.mark.gpu
def test_cpp_wrap_pair_giraf_equals_python_mirror(gpu_lib, tmp_path):
    """WrapPairGirafBatch of the C++ mirror == the Python mirror (both against the restatement in test_gsw_reads.py)"""
    _build()
    seqs, edges, reads = make_case(11, "wide")
    sc = MX["HumanChimpTwo"]
    write_case(str(tmp_path / "case.txt"), seqs, edges, reads, 16, 1, sc)
    assert subprocess.call([BIN, str(tmp_path / "case.txt"), str(tmp_path / "out.txt"), "pairs"]) == 0
    rows, _ = read_out(str(tmp_path / "out.txt"))
    g = build(seqs, edges)
    index = gg.SeedIndex(g.Nodes, 16, 1)
    pairs = [(gg.FastqBig("a", reads[2 * k]), gg.FastqBig("b", reads[2 * k + 1])) for k in range(len(reads) // 2)]
    py = gg.WrapPairGirafBatch(g, pairs, index, 16, sc)
    for k, (fw, rv) in enumerate(py):
        for r, row in ((fw, rows[2 * k]), (rv, rows[2 * k + 1])):
            key = r.key()
            assert row == key[:8] + (len(key[8]), r.Flag), "pair %d" % k


def test_worker_pool_on_the_cpu(tmp_path):
    """GswPool / parallelFor of the C++ mirror (host code only, no device): every index once in both partitions, the lowest failing index's
    exception, concurrent callers, the worker cap, Go's append capacities"""
    exe = str(tmp_path / "pool_test.bin")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "pool_test.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "pool ok" in r.stdout, r.stdout + r.stderr
