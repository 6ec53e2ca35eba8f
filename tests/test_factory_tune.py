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
