"""Markdown summary of one profile-set directory written by tools/r03_final.sh <name> prof (the rows of DESIGN.md section 5).
    python tools/summarise_run.py gpurun_out/<name>"""
import json
import os
import sys

d = sys.argv[1]
PF_MS = 99.35e9 * 64 / 1e12  # PFLOP/s x ms of one 64-frame step (sum 2 H W Cin Cout 9 over the conv launches)
PEAK = 2.5


def line(name):
    p = os.path.join(d, name)
    return json.loads(open(p).readline()) if os.path.exists(p) and os.path.getsize(p) else None


def trace(name):
    p = os.path.join(d, name)
    if not os.path.exists(p):
        return None
    conv = up = 0.0
    n = None
    for l in open(p):
        if not l.startswith("| `"):
            continue
        c = [x.strip() for x in l.split("|")]
        if "stem16" in c[1]:
            n = int(c[2])
        if any(k in c[1] for k in ("conv3x3_dma", "convpair", "stem16")):
            conv += float(c[3])
        if "upsample2x" in c[1]:
            up += float(c[3])
    return n, conv / n, up / n


j = line("bench_line.json")
if j:
    r, s, l8, pv = j["roofline"], j["sustained"], j["literal_split_8_per_gpu"], j["cpu_baseline"]["parity_vs_oracle"]
    print(f"| bench line (default `bench.py`) | **{j['value'] / 1e3:.2f} k frames/s, {j['ms_per_step']:.3f} ms per 64-frame step**; sustained "
          f"{s['steps_effective']} steps in {s['seconds']:.1f} s: {s['value'] / 1e3:.2f} k; 8 frames per GPU per step: {l8['ms_per_step']:.3f} ms = "
          f"{l8['value'] / 1e3:.2f} k frames/s per GPU |")
    print(f"| HIP events of the same run | network forward {r['network_ms_per_step']:.3f} ms: conv launches {r['frac']:.3f}, **whole forward "
          f"{r['frac_forward']:.3f} of 2.5 PFLOP/s**; every upsampling materialised: {r['frac_materialised']:.3f} / {r['frac_forward_materialised']:.3f}; "
          f"random-init (dense) weights: {r['frac_dense']:.3f} / {r['frac_forward_dense']:.3f} |")
    print(f"| parity block of the bench line ({pv['frames']} frames, {pv['peaks']} peaks) | {pv['peaks_within_0.5px']} within 0.5 px, max "
          f"{pv['max_peak_delta_px']} px, mean {pv['mean_peak_delta_px']} px; frames with a different count / assignment: "
          f"{pv['frames_with_different_instance_count']} / {pv['frames_with_different_node_assignment']} |")
    print(f"| CPU oracle (\"port\") | {j['cpu_baseline']['value']:.1f} frames/s on {j['cpu_baseline']['cores']} threads |")
t = trace("kt_kernel_stats.md")
if t:
    n, conv, up = t
    print(f"| rocprofv3 kernel trace ({n} forward passes) | conv launches {conv:.3f} ms/step = {PF_MS / conv / PEAK:.3f}; with the upsampling launches "
          f"{conv + up:.3f} ms = **{PF_MS / (conv + up) / PEAK:.3f} of peak over the whole forward** |")
p = os.path.join(d, "pmc_mfma_util.md")
if os.path.exists(p):
    print("| `SQ_VALU_MFMA_BUSY_CYCLES` | " + open(p).read().strip().split("\n")[-1] + " |")
p = os.path.join(d, "pmc_hbm_traffic.json")
if os.path.exists(p):
    t = json.load(open(p))
    print(f"| HBM traffic of the conv launches (FETCH x 2 + WRITE) | {t['conv_family_bytes_per_step'] / 1e9:.2f} GB per step "
          f"(fetch {t['fetch_x2_bytes_per_step'] / 1e9:.2f}, write {t['write_bytes_per_step'] / 1e9:.2f}; passes {t['forward_passes']}) |")
