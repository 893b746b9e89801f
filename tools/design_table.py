"""The round's tables of DESIGN.md section 5 / 6, from the committed evidence:   python tools/design_table.py r05 [--fill]
profiles/<round>_bench_details.json = the details file of ONE default bench.py run (the driver's command line), profiles/<round>_bench_n1.json its
final line, profiles/hbm_traffic.json = the rocprofv3 passes of tools/profile_cfg.sh.  --fill rewrites the marked blocks of DESIGN.md."""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r05"
det = json.load(open(os.path.join(root, "profiles", f"{tag}_bench_details.json")))
t = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))

rows = {}
for e in det.get("per_config", []) + [det["headline"]]:
    rows[(e["matrix"], e["impl"])] = e
for e in det.get("bm_list", []):
    rows.setdefault((e["matrix"], e["impl"]), e)
for e in det.get("bm_list_float", []):
    rows.setdefault((e["matrix"], e["impl"]), e)


def pct(v):
    return "—" if v is None else f"{v * 100:.1f}"


def line(key):
    e = rows[key]
    r = e.get("roofline", e)
    frac, step, cold = r["frac"], e.get("frac_whole_step", r.get("frac_whole_step")), r.get("frac_mall_cold")
    kernel_us, step_us = r["kernel_ms"] * 1e3, e["ms_per_step"] * 1e3
    steady = f" (steady {r['kernel_ms_steady'] * 1e3:.1f} µs = {pct(r['frac_steady'])} %)" if r.get("kernel_ms_steady") else ""      # round 6: `frac` = the all-launch average
    p = t.get(key[0])
    if p is not None and p.get("impl") is not None and int(p["impl"]) != ["fixed", "float_pob", "float_stall"].index(key[1]):
        p = None      # (profiled in another numeric mode)
    prof = traffic = "—"
    if p and p.get("round") == tag:
        prof = f"{p['kernel_avg_us']:.1f} / {p.get('kernel_steady_median_us', p['kernel_avg_us']):.1f} µs = {p['roofline_frac_rocprof'] * 100:.1f} %"
        traffic = f"{p['hbm_bytes_per_launch'] / 1e6:.0f} vs {8 * p['nnz'] / 1e6:.0f} MB ({p['hbm_bytes_per_launch'] / (8.0 * p['nnz']):.2f} ×)"
    fmt = e["stream_format"] + (f", {e['col_slices']} slices" if e.get("col_slices", 1) > 1 else "")
    parity = e["parity_vs_oracle"]
    parity = "bit-exact" if parity.startswith("bit-exact") else ("tolerance" + (" (csim abs 1e-4: " + parity.split("not met on ")[1].split(";")[0] + " over)" if "not met on" in parity else ""))
    long_run = ""
    if e.get("ms_per_step_long_run"):      # K = 20 steps of a small matrix carry the final synchronisation: the same loop over thousands of steps beside it
        lus = e["ms_per_step_long_run"] * 1e3
        long_run = f" ({lus:.1f} µs = {8.0 * e['nnz'] / (lus * 1e-6) / 8e12 * 100:.1f} % over {e['long_run_steps']} steps)"
    return f"| {key[0]} / {key[1]} | {fmt} | {kernel_us:.1f} µs = {pct(frac)} %{steady} | {step_us:.1f} µs = {pct(step)} %{long_run} | {pct(cold)} | {prof} | {traffic} | {parity} |"


order = [("ogbl_ppa", "fixed"), ("transformer_50", "float_pob"), ("ogbn_products", "float_stall"), ("mouse_gene", "fixed"), ("ogbl_ppa_rmat", "fixed"),
         ("gplus", "fixed"), ("hollywood", "fixed"), ("pokec", "fixed"), ("ogbn_products", "fixed"), ("transformer_50", "fixed"), ("transformer_60", "fixed"),
         ("transformer_70", "fixed"), ("transformer_80", "fixed"), ("transformer_90", "fixed"), ("transformer_95", "fixed"),
         ("transformer_80", "float_pob"), ("transformer_80", "float_stall"), ("mouse_gene", "float_pob"), ("mouse_gene", "float_stall"),
         ("pokec", "float_pob"), ("pokec", "float_stall"), ("ogbn_products", "float_pob")]
table = [f"**Round {int(tag[1:])}** (`profiles/{tag}_bench_n1.json` + `profiles/{tag}_bench_details.json` = ONE default `bench.py --gpus 1 --steps 20 --warmup 5` run, the driver's command line, one box, "
         f"every row checked against the oracle in the same run; rocprofv3 column and HBM traffic: `profiles/{tag}_<config>_rocprofv3_summary.txt`, `profiles/hbm_traffic.json`, "
         "another box of the same build — boxes differ by ± 2–3 %; one box in ~45 ran everything 8–35 % slower, `profiles/r05_slow_box_note.txt`; since round 6 the kernel column is the average over ALL launches of regions entered from an idle stream — what `rocprofv3 --stats` prints — with the steady state beside it).  The first five rows are BASELINE.json's configurations (+ the R-MAT stand-in), the rest the reference's sweep `sw/bm.sh` in all numeric modes:",
         "",
         "| matrix / IMPL | image | kernel alone (`roofline.frac`) | whole step (`value`) | whole step, MALL-cold % | rocprofv3 kernel avg / steady | HBM traffic vs 8·nnz | parity |",
         "|---|---|---|---|---|---|---|---|"]
table += [line(k) for k in order if k in rows]
table_text = "\n".join(table)

scal = {}
for s in det.get("strong_scaling_prediction", []):
    name = s["workload"].split(",")[0]
    scal[name] = "; ".join(f"{sp['n_gpus']} GPUs: slowest slab {sp['max_slab_us']:.1f} µs → {sp['predicted_compute_only_efficiency'] * 100:.0f} %"
                           + (f" (graph replay {sp['max_slab_us_graph']:.1f} µs → {sp['predicted_compute_only_efficiency_graph'] * 100:.0f} %)" if "max_slab_us_graph" in sp else "")
                           for sp in s["splits"]) + f" against {s['unsplit_us']:.1f} µs unsplit"

if "--fill" in sys.argv:
    path = os.path.join(root, "DESIGN.md")
    text = open(path).read()

    def put(marker, body):
        global text
        a, b = f"<!-- {marker} -->", f"<!-- /{marker} -->"
        if a in text:
            text = re.sub(re.escape(a) + r".*?" + re.escape(b), lambda m: a + "\n" + body + "\n" + b, text, flags=re.S)
        else:
            text = text.replace(marker.upper().replace("-", "_"), a + "\n" + body + "\n" + b)

    put("bench-table" if "<!-- bench-table -->" in text else "round5-table", table_text)
    for name, marker in (("mouse_gene", "scaling-mouse"), ("hollywood", "scaling-hollywood"), ("ogbn_products", "scaling-ogbn")):
        if name in scal:
            a, b = f"<!-- {marker} -->", f"<!-- /{marker} -->"
            if a in text:
                text = re.sub(re.escape(a) + r".*?" + re.escape(b), lambda m: a + scal[name] + b, text, flags=re.S)
            else:
                text = text.replace(marker.upper().replace("-", "_"), a + scal[name] + b)
    open(path, "w").write(text)
    print("DESIGN.md filled")
else:
    print(table_text)
    for k, v in scal.items():
        print(k, ":", v)
