#!/usr/bin/env python3
"""Copy what scripts/gpu_round4_profiles.sh left under gpurun_out/ into profiles/r04_* (the names DESIGN.md section 5 cites)."""
import json, shutil, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import bench
d = json.load(open("gpurun_out/traffic_workloads.json"))
print("traffic source_hash", d["source_hash"], "HEAD csrc", bench.source_hash())
pairs = [("traffic_workloads.json", "traffic_latest.json"), ("traffic_workloads.json", "r04_q_hbm_traffic_pmc.json"),
         ("prof_r04_headline/trace_kernel_stats.csv", "r04_r_headline_rotating_kernel_stats.csv"),
         ("prof_r04_b4096/trace_kernel_stats.csv", "r04_s_b4096_kernel_stats.csv"),
         ("prof_r04_config5/trace_kernel_stats.csv", "r04_t_config5_bf16_nhwc_kernel_stats.csv"),
         ("prof_r04_bigmaps/trace_kernel_stats.csv", "r04_u_bigmaps_kernel_stats_all_shapes.csv"),
         ("r04_fwd_pmc.csv", "r04_v_fwd_band_pmc.csv"),
         ("r04_bench_20.json", "r04_w_bench_line_steps20.json"), ("r04_bench_200.json", "r04_w_bench_line_steps200.json"),
         ("r04_bigmaps_final.jsonl", "r04_x_bigmaps_tile_kernels.jsonl"), ("r04_shape_sweep.jsonl", "r04_y_shape_sweep.jsonl"),
         ("r04_fused_callers_final.jsonl", "r04_z_fused_callers.jsonl"), ("r04_all_measures_final.jsonl", "r04_all_measures.jsonl"),
         ("r04_train.jsonl", "r04_zz_train_step_1gpu.jsonl")]
for a, b in pairs:
    src = os.path.join("gpurun_out", a)
    if not os.path.exists(src):
        print("MISSING", src)
        continue
    if a.endswith(".jsonl") or a.endswith(".json"):
        txt = "".join(l for l in open(src) if "amdgpu.ids" not in l)
        open(os.path.join("profiles", b), "w").write(txt)
    else:
        shutil.copy(src, os.path.join("profiles", b))
out = subprocess.run([sys.executable, "scripts/summarize_bigmaps_trace.py", "gpurun_out/prof_r04_bigmaps/trace_kernel_trace.csv"],
                     capture_output=True, text=True).stdout
open("profiles/r04_u2_bigmaps_rocprof_per_shape.csv", "w").write(out)
print(out)
