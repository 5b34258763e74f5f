"""CPU tier: the host logic of libditherpie_hip.so (dither_pie_amd/csrc/host_logic.h: scipy-order KD-tree build,
accelerator table assembly, diffusion candidate tables with their multi-threaded per-cell loops) compiled WITHOUT HIP
behind dither_pie_amd/csrc/host_sanitize.cpp, once with -fsanitize=address,undefined and once with -fsanitize=thread
(`make -C dither_pie_amd/csrc host_asan host_tsan`), and run on the golden cKDTree palettes.  A sanitizer report makes
the process exit non-zero (halt_on_error), a failed self-check exits 1."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "dither_pie_amd", "csrc")
ENV = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1:exitcode=66")


@pytest.fixture(scope="module")
def tools():
    subprocess.check_call(["make", "-s", "-C", CSRC, "host_asan", "host_tsan"])
    return {k: os.path.join(CSRC, "build", "host_" + k) for k in ("asan", "tsan")}


def _run(tool, cmd, pts, tmp_path, *extra):
    f = tmp_path / "pts.f64"
    np.ascontiguousarray(pts, dtype=np.float64).tofile(f)
    r = subprocess.run([tool, cmd, str(f), str(len(pts)), *map(str, extra)], capture_output=True, text=True, env=ENV,
                       timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r.stdout


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_kdtree_build_under_sanitizers_matches_scipy(tools, gold, kat, tmp_path, san):
    """build_tree (libstdc++ nth_element on raw index arithmetic) on all 11 golden palettes, incl. duplicates, a flat one
    and a float one: same indices / nodes / splits as scipy's cKDTree, nothing for ASan / UBSan / TSan to report."""
    for nm in kat["misc"]["tree_palettes"]:
        pts = gold[f"tree_{nm}_pts"]
        lines = _run(tools[san], "kdtree", pts, tmp_path).strip().split("\n")
        assert lines[0].split()[0] == "indices"
        assert np.array_equal(np.array(lines[0].split()[1:], np.int32), gold[f"tree_{nm}_indices"]), nm
        n = int(lines[1].split()[1])
        rows = [l.split() for l in lines[2:2 + n]]
        nodes = np.array([[int(v) for v in r[:5]] for r in rows], np.int32)
        splits = np.array([float(r[5]) for r in rows])
        assert np.array_equal(nodes, gold[f"tree_{nm}_nodes"]), nm
        inner = nodes[:, 0] >= 0
        assert np.array_equal(splits[inner], gold[f"tree_{nm}_splits"][inner]), nm


def _crowded(K, seed):
    rs = np.random.RandomState(seed)
    return np.clip(np.round(40 + rs.randn(K, 3) * 9), 0, 255)


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_diffusion_tables_under_sanitizers(tools, gold, tmp_path, san):
    """ed_tables_refine with its up-to-8-thread per-cell loops: random / uniform-grid / float (gamma) / crowded palettes
    (the crowded ones overflow cells and go through the octree refinement); the harness checks on 40 000 points that the
    true nearest entries are on every list a kernel would search."""
    wide = np.random.RandomState(5).randint(0, 256, (700, 3)).astype(np.float64)   # 257..1024 colours: ten-bit list entries
    cases = [gold["tree_p256_pts"], gold["tree_p32_pts"], gold["tree_p11_pts"], gold["tree_U16_pts"], gold["tree_lin256_pts"],
             gold["tree_dup40_pts"], _crowded(200, 1), _crowded(16, 2), wide, _crowded(300, 3)]
    if san == "tsan":
        cases = [cases[0], cases[3], cases[6], cases[9]]
    for pts in cases:
        out = _run(tools[san], "edtables", pts, tmp_path)
        assert " bad=0" in out, out


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_accelerator_table_assembly_under_sanitizers(tools, gold, tmp_path, san):
    """assemble_table / crowded_nodes_first / make_warp / mass_points / entries_in_split_cells / compact_table from brute-force
    membership masks (the harness scans all 2^24 colours as accel_scan_kernel does): a random, a uniform-grid and a
    crowded palette, 8- and 4-entry blocks; every sampled colour finds all of T(x) in the block it reaches."""
    cases = [(gold["tree_p32_pts"], 8), (gold["tree_U16_pts"], 4), (_crowded(40, 3), 8)]
    if san == "tsan":
        cases = cases[2:]
    for pts, bw in cases:
        out = _run(tools[san], "accel", pts, tmp_path, bw)
        assert " bad=0 " in out and "warp_bad=0" in out and "compact_bad=0" in out, out


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_median_cut_replay_under_sanitizers(tools, tmp_path, san):
    """pyset_order (the replay of CPython's set: hash, probing, growth) and median_cut_rgb (counting sorts, recursion on
    raw pointers) under ASan / UBSan / TSan: the order equals this interpreter's `list(set(...))`, the palette equals the
    Python cut -- 60 000 colours (past the change of growth policy), a handful of colours with depth 8 (empty buckets), one
    colour, many duplicates."""
    from dither_pie_amd.dithering_lib import ColorReducer
    rs = np.random.RandomState(9)
    cases = [(rs.randint(0, 256, (60000, 3)), 8), (rs.randint(0, 6, (5000, 3)), 8), (np.full((100, 3), 7), 3), (rs.randint(0, 256, (1, 3)), 0),
             (rs.randint(100, 140, (30000, 3)), 5)]
    if san == "tsan":
        cases = cases[1:3]
    for arr, depth in cases:
        arr = np.ascontiguousarray(arr.astype(np.uint8))
        f = tmp_path / "rgb.u8"
        arr.tofile(f)
        r = subprocess.run([tools[san], "mediancut", str(f), str(len(arr)), str(depth)], capture_output=True, text=True, env=ENV, timeout=600)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        lines = r.stdout.strip().split("\n")
        real = list(set(zip(arr[:, 0].tolist(), arr[:, 1].tolist(), arr[:, 2].tolist())))
        assert int(lines[0].split()[1]) == len(real)
        order = np.array(lines[2].split()[1:], np.int64)
        assert [tuple(int(v) for v in c) for c in arr[order]] == real[:len(order)]
        pal = np.array(lines[1].split()[1:], np.int64).reshape(-1, 3)
        ref = ColorReducer._median_cut_arrays(np.array(real, np.uint8).reshape(-1, 3), depth)
        assert [tuple(c) for c in pal.tolist()] == [tuple(int(v) for v in c) for c in ref]
