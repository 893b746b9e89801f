"""The round's roofline table of DESIGN.md section 5, from the committed evidence: python tools/design_table.py r04
(profiles/<round>_bench_n1.json = one bench.py run; profiles/hbm_traffic.json = the rocprofv3 passes of tools/profile_cfg.sh)."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = json.loads(open(os.path.join(root, "profiles", f"{tag}_bench_n1.json")).read().strip().splitlines()[-1])
t = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
bm = {e["matrix"]: e for e in d["bm_list"]}


def line(label, key, e=None, bm_entry=None):
    p = t[key]
    prof = f"{p['kernel_avg_us']:.1f} / {p.get('kernel_steady_median_us', p['kernel_avg_us']):.1f} µs"
    traffic = f"{p['hbm_bytes_per_launch'] / 1e6:.0f} MB vs {8 * p['nnz'] / 1e6:.0f} MB ({p['hbm_bytes_per_launch'] / (8.0 * p['nnz']):.2f} ×)"
    if e is not None:
        r = e["roofline"]
        whole = e.get("hbm_roofline_fraction_whole_job", d["hbm_roofline_fraction_whole_job"])
        return (f"| {label} | {prof} | {r['kernel_ms'] * 1e3:.1f} µs | {e['ms_per_step'] * 1e3:.1f} µs | {r['mall_cold']['ms_per_step_round_robin'] * 1e3:.1f} µs | "
                f"{p['roofline_frac_rocprof'] * 100:.1f} / {r['frac'] * 100:.1f} / {whole * 100:.1f} / {r['frac_mall_cold'] * 100:.1f} | {p['step_us_wall_best']:.1f} µs | {traffic} |")
    b = bm_entry
    return (f"| {label} | {prof} | {b['kernel_ms'] * 1e3:.1f} µs | {b['ms_per_step'] * 1e3:.1f} µs | — | "
            f"{p['roofline_frac_rocprof'] * 100:.1f} / {b['frac_kernel'] * 100:.1f} / {b['hbm_roofline_fraction_whole_job'] * 100:.1f} / — | {p['step_us_wall_best']:.1f} µs | {traffic} |")


print("| configuration (format) | kernel, rocprofv3 avg / steady | kernel, `bench.py` | whole job, warm | whole job, MALL-cold | % of 8 TB/s: rocprof / kernel / whole / cold | step, unprofiled, same process as the profile | HBM traffic per launch vs 8·nnz |")
print("|---|---|---|---|---|---|---|---|")
print(line("ogbl-ppa, fixed (DELTA, 4 slices)", "ogbl_ppa", d))
for (label, key), e in zip([("transformer-50, float_pob (BITMAP)", "transformer_50"), ("ogbn-products, float_stall (OWNER24, 5 slices)", "ogbn_products"),
                            ("mouse_gene, fixed (DELTA + lane sums)", "mouse_gene"), ("ogbl-ppa R-MAT stand-in, fixed (PAIRS, 4 slices)", "ogbl_ppa_rmat")], d["per_config"]):
    print(line(label, key, e))
for label, key in (("pokec, fixed (SWEEP)", "pokec"), ("hollywood, fixed (DELTA, 2 slices)", "hollywood"), ("gplus, fixed (DELTA, 7 slices)", "gplus")):
    print(line(label, key, bm_entry=bm[key]))
