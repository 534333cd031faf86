#!/usr/bin/env python3
"""Short view of a bench.py JSON line: tools/print_bench.py <file>"""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f maps/s  %.3f ms/step" % (j["value"], j["ms_per_step"]))
r = j.get("roofline", {})
print("K3 %.3f ms  %.0f GB/s  frac %.3f  of stream peak %.3f (%.0f GB/s)" % (r.get("avg_launch_ms", 0), r.get("achieved", 0), r.get("frac", 0),
      r.get("frac_of_measured_stream", 0), r.get("measured_stream_peak_gbs", 0)))
for k in ("pipelined", "h2d_inclusive", "conv0_fp32_mfma", "exact_grid", "roofline_mfma", "path_a", "path_a_pipelined", "path_a_vendor_convs", "path_a_half_dispnet", "cpu_baseline"):
    v = j.get(k)
    if v:
        print(k, {kk: (round(v[kk], 4) if isinstance(v[kk], float) else v[kk]) for kk in v
                  if kk in ("value", "ms_per_step", "max_rel_depth_diff_vs_headline_model", "achieved", "frac", "avg_launch_ms",
                            "warp_variance_ms", "error", "sweep_corr_ms", "skipped", "max_rel_depth_diff_vs_path_a")})
