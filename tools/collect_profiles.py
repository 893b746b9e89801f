"""Copy the per-config rocprofv3 summaries of tools/profile_cfg.sh (gpurun_out/prof_<config>/) into profiles/ and merge their
counter results into profiles/hbm_traffic.json (keyed by config; bench.py reads roofline.traffic from there):
python tools/collect_profiles.py <round tag, e.g. r02> [configs...]"""
import hashlib, json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():      # == bench.py: sources_sha16()
    h = hashlib.sha256()
    csrc = os.path.join(root, "hisparse_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


try:
    git_head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "hisparse_amd/csrc"], capture_output=True, text=True).stdout.strip())
except OSError:
    git_head, dirty = None, None
tag = sys.argv[1]
configs = sys.argv[2:] or ["ogbl_ppa", "transformer_50", "ogbn_products", "mouse_gene", "ogbl_ppa_rmat", "pokec", "hollywood", "gplus", "transformer_80", "transformer_95"]
path = os.path.join(root, "profiles", "hbm_traffic.json")
try:
    merged = json.load(open(path))
    if "hbm_bytes_per_launch" in merged:      # round-1 layout: one flat entry for ogbl-ppa
        merged = {}
except (OSError, ValueError):
    merged = {}
for c in configs:
    src = os.path.join(root, "gpurun_out", f"prof_{c}")
    if not os.path.exists(os.path.join(src, "summary.txt")):
        print("missing", src)
        continue
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(root, "profiles", f"{tag}_{c}_rocprofv3_summary.txt"))
    merged[c] = json.load(open(os.path.join(src, "hbm_traffic.json")))
    e = merged[c]
    if "step_us_wall_best" in e and "kernel_avg_us" in e:      # (re-stated here so that summaries taken with an older summariser carry the same verdict)
        steady = e.get("kernel_steady_median_us", e["kernel_avg_us"])      # tools/summarize_profile.py: median of the trace's steady half
        e["kernel_fits_inside_the_timed_step"] = steady <= e["step_us_wall_best"] * 1.01 + 0.6
        e["kernels_sum_minus_step_us"] = e.get("step_kernels_avg_us", e["kernel_avg_us"]) - e["step_us_wall_best"]
        with open(os.path.join(root, "profiles", f"{tag}_{c}_rocprofv3_summary.txt"), "a") as f:
            f.write(f"verdict: dominant kernel, steady {steady:.2f} us (all-dispatch average {e['kernel_avg_us']:.2f}) <= unprofiled step {e['step_us_wall_best']:.2f} us (1 % + 0.6 us of "
                    f"profiler cost per dispatch allowed): {'yes' if e['kernel_fits_inside_the_timed_step'] else 'NO'}; kernels' sum under the profiler minus the step: {e['kernels_sum_minus_step_us']:+.2f} us\n")
    merged[c]["round"] = tag
    # which build the counters were taken on: run this right after the gpurun call, before touching the sources again
    merged[c]["git_head"] = (git_head + ("+uncommitted changes" if dirty else "")) if git_head else None
    merged[c]["csrc_sha16"] = sources_sha16()
    print(c, "kernel avg us", merged[c].get("kernel_avg_us"), "HBM bytes", merged[c].get("hbm_bytes_per_launch"), "frac", merged[c].get("roofline_frac_rocprof"))
json.dump(merged, open(path, "w"), indent=1)
