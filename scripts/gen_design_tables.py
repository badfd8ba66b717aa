"""The kernel table of DESIGN.md section 5, generated from a bench line (profiles/r06_bench.json by default):
    python scripts/gen_design_tables.py [bench.json] [--write]     (--write replaces the block between the kernel-table markers)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND = {"k_msm_accumulate_fb": "VALU issue (7-multiplication mixed addition x 16 windows) co-bound with random 128-byte gathers",
         "k_fold_gens_tab": "VALU (~160 mixed additions + 33 doublings per output; the kind's kernel is k_fold_gens_w, the event-list fold, on the fixed generators' table)", "k_fold_gens": "VALU at 2 waves/SIMD (segmented chains)",
         "k_msm_bin_l1+l2 / k_msm_scatter_lds": "LDS atomics + HBM writes (no field arithmetic)", "k_msm_reduce_level+fused": "latency: three dependent launches, the last on few blocks",
         "k_msm_small": "latency: one block per (problem, window)", "k_msm_accumulate_gen": "VALU / latency", "other": "mixed (k_ipp_round, k_poly_t, k_lr_vec, k_nonce_expand, ...)",
         "k_decode / k_commit": "VALU (inverse-square-root chains)", "k_verify_scalars": "scalar-field arithmetic", "k_sigma_prove / k_sigma_vprep / k_sigma_verify": "VALU"}


def table(path):
    j = json.loads(open(path).read().strip().splitlines()[-1])
    k = j["kernels"]; peak = k["fe_mul_per_s_peak_measured"]
    rows = ["| Kernel (kind) | launches / client | ms / client | avg launch ms | algorithmic GB/s (frac of 8 TB/s) | frac of the Fp-mul ceiling | bound |", "|---|---|---|---|---|---|---|"]
    for t in k["top"]:
        ms = t.get("ms_per_client", t.get("ms_per_unit")); ln = t.get("launches_per_client", t.get("launches_per_unit"))
        rows.append("| `%s` | %.1f | %.2f | %.3f | %.0f (%.4f) | %s | %s |" % (t["kernel"], ln, ms, t["avg_launch_ms"], t["algorithmic_GBps"], t["hbm_frac"],
                                                                         ("%.2f" % t["fe_mul_frac_of_peak"]) if t.get("fe_mul_frac_of_peak") else "—", BOUND.get(t["kernel"], "")))
    r, v = j["roofline"], j["valu_roofline"]
    head = ("`%s` (BASELINE cfg 2, one client per step; HIP events of %s instrumented steps; ceiling %.3g field multiplications/s measured in the same process):\n\n"
            % (os.path.basename(path), k.get("instrumented_steps", "the"), peak))
    tail = ("\n\n`roofline` block of that line: kernel `%s`, algorithmic %.1f MB per launch in %.3f ms = **%.1f GB/s = frac %.4f of 8 TB/s**; PMC traffic %.2f GB per launch "
            "(`traffic_frac` %.2f; layout minimum %.2f); scalar + point accounting %.1f MB = frac %.4f; `valu_roofline.frac` **%.2f**, `issue.issue_frac` %s, `end_to_end_frac` %.2f."
            % (r["kernel"], r["algorithmic_bytes_per_launch"] / 1e6, r["avg_launch_ms"], r["achieved"], r["frac"], (r["traffic"] or 0) / 1e9, r["traffic_frac"] or 0,
               r.get("layout_min_frac") or 0, (r.get("algorithmic_bytes_per_launch_scalar_and_point") or 0) / 1e6, r.get("frac_scalar_and_point") or 0, v["frac"],
               ("%.2f" % v["issue"]["issue_frac"]) if (v.get("issue") or {}).get("issue_frac") else "—", v["end_to_end_frac"]))
    return head + "\n".join(rows) + tail


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "profiles", "r06_bench.json")
    t = table(path)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md"); d = open(p).read()
        a, b = d.index("<!-- kernel-table:begin -->") + len("<!-- kernel-table:begin -->"), d.index("<!-- kernel-table:end -->")
        open(p, "w").write(d[:a] + "\n" + t + "\n" + d[b:])
    else:
        print(t)
