"""Register / scratch budgets of the wavefront kernels, read from the gfx950 code object inside the built library (CPU suite: no GPU
needed).  The kernels are paced by waves per SIMD; a change that silently pushes one over its budget (round 3: a run-time choice
between two beta encodings inside the unrolled steps took fill_affine_kernel from 152 to 512 registers plus scratch, one wave per
SIMD, 1.8x slower) shows up here instead of in a profile several commits later."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp_path):
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "gonomics_amd", "libgonomics_align_hip.so")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
        if "name" not in f or ".kd" in f["name"]:
            continue
        out[f["name"]] = {k: int(f[k]) for k in ("agpr_count", "vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count")}
    names = sorted(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    res = {}
    for n, d in zip(names, dem):
        d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "")
        res[d] = out[n]
    return res


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(f"{LLVM}/llvm-readelf"):
        pytest.skip("no llvm-readelf")
    return _kernels(tmp_path_factory.mktemp("co"))


def _waves_per_simd(k):
    regs = k["vgpr_count"] + k["agpr_count"]            # unified register file of 512 per SIMD lane, granule 8
    return min(8, 512 // max(8, (regs + 7) // 8 * 8))


# kernel (prefix of the demangled name) -> waves per SIMD its register count must allow
BUDGET = [
    ("fp_sweep_kernel<19, false", 3), ("fp_sweep_kernel<20, false", 3), ("fp_sweep_kernel<19, true", 3), ("fp_sweep_kernel<20, true", 3), ("fp_sweep_levels_kernel", 2),
    ("fill_affine_kernel<false, false", 3), ("fill_affine_kernel<true, false", 3), ("fill_affine_kernel<false, true", 2), ("fill_affine_kernel<true, true", 2),
    ("fill_const_kernel<false, 0", 3), ("fill_const_kernel<false, 1", 3), ("fill_const_kernel<false, 2", 3), ("fill_const_kernel<true", 2),
    ("cl_sweep_kernel<true, false>", 5), ("cl_sweep_wg_kernel<4>", 5), ("cl_sweep_flat_kernel<true, false>", 4), ("cl_sweep_flat_kernel<false, false>", 3), ("cl_sweep_kernel<false, false>", 3),
    ("al_sweep_kernel<true, false>", 3), ("al_sweep_kernel<false, false>", 3),
    # the REBASE instantiations (pairs beyond the static int32 range, round 5) carry an int64 base and two deltas: one wave per SIMD less is their price
    ("cl_sweep_kernel<true, true>", 4), ("cl_sweep_flat_kernel<true, true>", 3), ("cl_sweep_kernel<false, true>", 3), ("cl_sweep_flat_kernel<false, true>", 3), ("al_sweep_kernel<true, true>", 3), ("al_sweep_kernel<false, true>", 2),
    ("fp_walk_kernel", 5), ("traceback_kernel", 4), ("gsw_traceback_kernel", 8),
    # the window walk with one LANE per pair (batches of more than 32 768 reads of several row blocks): 98 registers since it carries the block
    # shortcut of round 4 (score table in LDS, two row-buffer keys, the diagonal's sum) -- a launch of <= 2 waves per SIMD, whatever its registers allow
    ("fp_walk_kernel<false, false, false, false>", 4),
    # rounds 5 / 6: the latency geometry, the int64 kernel, the 64-lane snapshot kernels (rows per lane 6 / 8 / 10 / 16: the sweep of ONE long pair is
    # strips(RW) waves on 1 024 SIMDs, up to 2 605 of them at RW = 6 -- three per SIMD must fit) and the walk farm
    ("lat_fill_kernel<false, false, false, false>", 8), ("lat_fill_kernel<true, false, false, false>", 6), ("lat_fill_kernel<true, true, false, false>", 5),
    ("lat_fill_kernel<true, false, true, false>", 3), ("lat_fill_kernel<true, false, true, true>", 4), ("lat_wide_kernel<false", 8), ("lat_wide_kernel<true", 5), ("lat_wide_kernel<true, false, true>", 3),
    ("al64_sweep_kernel<6, true>", 5), ("al64_sweep_kernel<6, false>", 4), ("al64_sweep_kernel<8, true>", 4), ("al64_sweep_kernel<8, false>", 4),
    ("al64_sweep_kernel<10, true>", 4), ("al64_sweep_kernel<10, false>", 3), ("al64_sweep_kernel<16, true>", 3), ("al64_sweep_kernel<16, false>", 2),
    ("cl64_sweep_kernel<10, true>", 5), ("cl64_sweep_kernel<10, false>", 4), ("cl64_sweep_kernel<4", 7),
    ("al64_farm_round_kernel<6", 3), ("al64_farm_round_kernel<8", 3), ("al64_farm_round_kernel<10", 2), ("al64_farm_round_kernel<16", 2), ("cl64_farm_round_kernel", 6),
    ("farm_walk_kernel<true", 3), ("farm_walk_kernel<false", 6), ("al64_walk_kernel", 2), ("al64_walk2_kernel", 3), ("cl64_walk_kernel", 4), ("cl64_walk2_kernel", 5),
]
# kernels that are allowed scratch (register-bound by design: their tiles live in LDS at one workgroup of 4 pairs per half CU)
SCRATCH_OK = ("al_walk_kernel",)
# a few values parked in scratch OUTSIDE the steady loops (the ISA listing shows the spills in the prologue / epilogue of the headline sweep -- it sits
# exactly at the 168 registers of three waves per SIMD -- and one reload per 16-step block of address pairs in the multi-strip constant-gap sweep)
SCRATCH_SMALL = {"fp_sweep_kernel<": 256, "cl_sweep_wg_kernel<": 32}
# LDS per workgroup: handed out in granules of 1280 B on gfx950 (160 KB per CU) -- the budgets are granule counts
LDS_GRANULES = [("fp_sweep_kernel", 11), ("fp_sweep_levels_kernel", 11), ("cl_sweep_kernel<true,", 6), ("cl_sweep_wg_kernel<4>", 25), ("cl_sweep_flat_kernel<true,", 6), ("fill_const_kernel<false, 0, true>", 6),
                ("fill_affine_kernel<false, false, false, true, false, false, false>", 11),
                # rounds 5 / 6: the sweeps of one long pair hold a wave's profile only; a round of the farm (walk window 52 KB + one re-fill's profile) and the
                # one-workgroup walks (one / two tiles of three direction planes in LDS) must stay inside the 64 KB / 160 KB a workgroup may declare
                ("al64_sweep_kernel<6, true>", 4), ("al64_sweep_kernel<8, true>", 5), ("al64_sweep_kernel<10, true>", 6), ("al64_sweep_kernel<16, true>", 9), ("al64_sweep_kernel<16, false>", 17),
                ("cl64_sweep_kernel<10, true>", 6), ("cl64_sweep_kernel<4, true>", 3), ("lat_fill_kernel", 3), ("lat_wide_kernel", 3),
                ("al64_farm_round_kernel", 57), ("cl64_farm_round_kernel", 21), ("farm_walk_kernel", 41), ("al64_walk_kernel", 67), ("al64_walk2_kernel", 125), ("cl64_walk2_kernel", 67)]


def test_every_wavefront_kernel_is_in_the_library(kernels):
    for prefix, _ in BUDGET:
        assert any(k.startswith(prefix) for k in kernels), prefix


def test_register_budgets(kernels):
    bad = []
    for name, k in kernels.items():
        match = [(len(prefix), waves) for prefix, waves in BUDGET if name.startswith(prefix)]
        if match and _waves_per_simd(k) < max(match)[1]:  # (the longest prefix decides)
            bad.append((name, k["vgpr_count"], k["agpr_count"], _waves_per_simd(k), max(match)[1]))
    assert not bad, bad


def test_no_scratch_outside_the_declared_kernels(kernels):
    bad = [(n, k["private_segment_fixed_size"], k["vgpr_spill_count"]) for n, k in kernels.items()
           if (k["private_segment_fixed_size"] or k["vgpr_spill_count"]) and not n.startswith(SCRATCH_OK) and "rocprim" not in n and "hipcub" not in n
           and not any(n.startswith(pre) and k["private_segment_fixed_size"] <= lim for pre, lim in SCRATCH_SMALL.items())]
    assert not bad, bad


def test_lds_budgets(kernels):
    bad = []
    for prefix, granules in LDS_GRANULES:
        for name, k in kernels.items():
            if name.startswith(prefix) and (k["group_segment_fixed_size"] + 1279) // 1280 > granules:
                bad.append((name, k["group_segment_fixed_size"], granules))
    assert not bad, bad
