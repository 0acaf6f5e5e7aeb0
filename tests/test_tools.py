"""The profile tooling that produces committed numbers (profiles/r<NN>_pmc_hbm_traffic.json, newest round, is what bench.py reads
`roofline.traffic` from): tools/pmc_traffic.py must count the forward passes of a profiled run itself -- bench.py warms up by
time, so the FETCH and the WRITE pass hold different numbers of launches."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, counter, passes):
    rows = []
    for _ in range(passes):
        rows.append(("void (anonymous namespace)::stem16_gray_kernel((anonymous namespace)::Stem16Params)", 33554432, 100 * 1024))
        for _ in range(2):  # two launches per pass of this kernel
            rows.append(("void (anonymous namespace)::conv3x3_dma_kernel<2, 16, 8, 2, 2, false, 0, false, false, 3>(x)", 1048576, 50 * 1024))
        rows.append(("void (anonymous namespace)::upsample2x_bilinear_c16_kernel<2>(x)", 4194304, 999 * 1024))  # not conv family
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        for name, grid, val in rows:
            w.writerow([name, grid, counter, val])
            w.writerow([name, grid, "OTHER", 1])


def test_pmc_traffic_counts_the_passes_of_each_run(tmp_path):
    fetch, write, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "t.json"
    _write(fetch, "FETCH_SIZE", 7)   # counters are in KB: 100 MB + 2 x 50 MB per pass, doubled for FETCH
    _write(write, "WRITE_SIZE", 11)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(fetch), str(write), "0", str(out)],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    t = json.load(open(out))
    assert t["forward_passes"] == [7, 11]
    assert abs(t["fetch_x2_bytes_per_step"] - 2 * 200 * 1048576) < 1 and abs(t["write_bytes_per_step"] - 200 * 1048576) < 1
    assert abs(t["conv_family_bytes_per_step"] - 600 * 1048576) < 1


def test_committed_traffic_file_is_what_bench_reads():
    sys.path.insert(0, ROOT)
    import bench

    import glob

    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))[-1]  # bench.py reads the newest round's
    assert os.path.basename(newest) >= "r04_pmc_hbm_traffic.json"
    t = json.load(open(newest))
    assert t["frames_per_step"] == 64 and t["size"] == 1024
    got = bench.profiled_traffic(64, 1024)
    assert got is not None and abs(got[0] - t["conv_family_bytes_per_step"]) < 1


def test_pers_store_count_model_matches_the_isa():
    """the PERS kernels' counted `s_waitcnt vmcnt(S)` rests on ONE global_store per valid 16-channel piece and output
    row; the ISA of the translation unit (cross-compiled here, ~30 s) must hold exactly that many store instructions per PERS
    instantiation, or a compiler change could let a tile start on an LDS stage whose copy is still in flight."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_pers_stores

    res = check_pers_stores.pers_store_counts()
    assert len(res) == 4
    for name, got, want in res:
        assert got == want, (name, got, want)
