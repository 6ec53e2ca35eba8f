"""The tile table shipped with the package (uniter_amd/tuned/gfx950.json) must belong to THIS build of the GEMM family:
the entries are tile indices, so a table left over from another tile list would silently select wrong (or illegal)
kernels.  Host-only checks — no GPU needed."""
import json
import os
import re

from uniter_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiles_in_source():
    src = open(os.path.join(ROOT, "uniter_amd", "csrc", "gemm.hip")).read()
    body = re.search(r"constexpr TileShape kTiles\[\] = \{(.*?)\};", src, re.S).group(1)
    return [tuple(int(v) for v in t) for t in re.findall(r"\{(\d+), (\d+), (\d+), (\d+)\}", body)]


def test_shipped_table_matches_the_build():
    table = json.load(open(ops.FACTORY_TUNE))
    tiles = _tiles_in_source()
    assert table["n_tiles"] == len(tiles) == _lib.C.uniter_gemm_tile_count()
    for e in table["gemm"]:
        assert 0 <= e["cfg"] < len(tiles) and e["splits"] >= 1
        bm, bn, _, ws = tiles[e["cfg"]]
        if e["kind"] == 1:                                   # dgrad: K-strided N-side operand, output columns K
            assert (bn in (64, 128, 192) or ws == 3) and e["K"] % bn == 0
        if e["kind"] == 0:
            assert e["N"] % bn == 0
        if e["kind"] == 3:                                   # grouped weight gradients: both operands K-strided
            assert (bm in (64, 128) and bn in (64, 128)) or ws == 3
            assert e["splits"] == 1 or (ws == 3 and e["splits"] == 2)   # two K slices: the eight-phase tile only
        if ws:
            contraction = {0: e["K"], 1: e["N"], 3: e["M"]}[e["kind"]]
            assert contraction % 64 == 0


def test_shipped_table_covers_the_benchmark_shape_and_installs():
    s = ops._shape(dict(H=768, heads=12, I=3072, p_hidden=0.1, p_attn=0.1, ln_eps=1e-12), 32, 96, True)
    table = json.load(open(ops.FACTORY_TUNE))
    have = {(e["kind"], e["M"], e["N"], e["K"]) for e in table["gemm"]}
    assert all(k in have for k in ops._layer_gemm_shapes(s))
    assert ops._load_tune_cache(ops.FACTORY_TUNE, s)                       # the library accepts every choice (host-side legality)
    small = ops._shape(dict(H=128, heads=2, I=256, p_hidden=0.0, p_attn=0.0, ln_eps=1e-12), 2, 32, True)
    assert not ops._load_tune_cache(ops.FACTORY_TUNE, small)               # other shapes fall through to the tuner


def test_committed_pmc_traffic_files_attribute_bytes_to_shapes():
    """bench.py fills roofline.traffic from profiles/*_pmc_traffic.json through its `by_shape` table ((epilogue, M, N, K) of
    the C-ABI call -> HBM bytes per launch).  A summary whose attribution step found nothing (round 2 shipped one: the passes
    had stopped writing the tile-choice file the summariser looked for) silently turns that field into null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    assert files, "no PMC traffic summary committed"
    for f in files:
        doc = json.load(open(f))
        shapes = doc.get("by_shape", {})
        assert shapes, "%s: empty by_shape" % os.path.basename(f)
        assert any(k.startswith("13:") for k in shapes), "%s: the grouped weight-gradient launch is missing" % os.path.basename(f)
        for k, e in shapes.items():
            parts = k.split(":")
            # (the deferred weight-gradient launch — kind 13 with more than one layer's four problems — runs once per step)
            need = 3 if parts[0] == "13" and int(parts[3]) > 4 else 8
            assert len(parts) == 4 and e["hbm_bytes"] > 0 and e["dispatches"] >= need, (f, k)


def test_summariser_attributes_synthetic_pmc_passes_to_every_shape(tmp_path):
    """scripts/summarize_profile.pmc_traffic on synthetic FETCH_SIZE / WRITE_SIZE passes built from the SHIPPED tile table:
    every encoder GEMM of the benchmark shape (kernel template + dispatch grid derived from the table, shapes that share a
    template told apart by program order) and the once-per-step deferred launch must come out under their bench.py keys."""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("summarize_profile", os.path.join(ROOT, "scripts", "summarize_profile.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    tiles = _tiles_in_source()
    table = json.load(open(ops.FACTORY_TUNE))
    H, I, M = 768, 3072, 3072
    lookup = {(e["kind"], e["N"], e["K"]): e for e in table["gemm"] if e["M"] == M and e["kind"] != 3 and min(e["N"], e["K"]) == H}
    order = [(0, 0, 3 * H, H), (0, 2, H, H), (0, 1, I, H), (0, 2, H, I), (1, 4, H, I), (1, 3, I, H), (1, 3, H, H), (1, 3, 3 * H, H)]
    launches = []                                             # (kernel name, grid size) of one layer, in program order
    for kind, epi, N, K in order:
        e = lookup[(kind, N, K)]
        bm, bn, st, ws = tiles[e["cfg"]]
        rows, cols = (M, N) if kind == 0 else (M, K)
        threads = 256 if ws == 0 else (512 if ws in (1, 3, 4) else 768)
        grid = ((rows + bm - 1) // bm) * (cols // bn) * threads * e["splits"]
        layout = {0: "false, false", 1: "false, true"}[kind]
        if ws >= 3:
            name = "void (anonymous namespace)::gemm%d_kernel<%s, %d>((anonymous namespace)::GemmArgs)" % (8 if ws == 3 else 6, layout, epi)
        else:
            name = "void (anonymous namespace)::gemm_kernel<%d, %d, %s, %d, %d, %d>((anonymous namespace)::GemmArgs)" % (bm, bn, layout, epi, st, ws)
        launches.append((name, grid, epi, N, K))
    multi = ("(anonymous namespace)::gemm8_multi_kernel((anonymous namespace)::GemmArgs const*, int const*, int, int, int, int)", 1600 * 512)

    def write_pass(path, counter, base):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            did = 0
            for step in range(5):
                for layer in range(12):
                    for j, (name, grid, epi, N, K) in enumerate(launches):
                        did += 1
                        w.writerow([did, name, grid, counter, base + 100.0 * j])      # KiB; distinct per position in the layer
                did += 1
                w.writerow([did, multi[0], multi[1], counter, 1000.0 * base])

    run = tmp_path / "prof" / "pmc_x" / "runc"
    run.mkdir(parents=True)
    fcsv, wcsv = str(run / "1_fetch.csv"), str(run / "1_write.csv")
    write_pass(fcsv, "FETCH_SIZE", 1000.0)
    write_pass(wcsv, "WRITE_SIZE", 500.0)
    dst = str(tmp_path / "out.json")
    sp.pmc_traffic(fcsv, wcsv, dst)
    shapes = json.load(open(dst))["by_shape"]
    for j, (name, grid, epi, N, K) in enumerate(launches):
        key = "%d:%d:%d:%d" % (epi, M, N, K)
        assert key in shapes, (key, sorted(shapes))
        assert shapes[key]["hbm_read_bytes"] == round(2.0 * (1000.0 + 100.0 * j) * 1024.0), key     # gfx950: FETCH_SIZE doubled
        assert shapes[key]["hbm_write_bytes"] == round((500.0 + 100.0 * j) * 1024.0), key
    key = "13:%d:%d:%d" % (M, 12 * (2 * H * I + 4 * H * H), 48)
    assert key in shapes and shapes[key]["dispatches"] == 5, sorted(shapes)

