#!/usr/bin/env python3
"""Copy what scripts/gpu_round3_profiles.sh left under gpurun_out/ into profiles/r03_* (the names DESIGN.md section 5 cites)."""
import json, shutil, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import bench
d = json.load(open("gpurun_out/traffic_workloads.json"))
print("traffic source_hash", d["source_hash"], "HEAD csrc", bench.source_hash())
pairs = [("traffic_workloads.json", "traffic_latest.json"), ("traffic_workloads.json", "r03_o_hbm_traffic_pmc.json"),
         ("prof_r03_headline/trace_kernel_stats.csv", "r03_k_headline_rotating_kernel_stats.csv"),
         ("prof_r03_b4096/trace_kernel_stats.csv", "r03_l_b4096_kernel_stats.csv"),
         ("prof_r03_config5/trace_kernel_stats.csv", "r03_m_config5_bf16_nhwc_kernel_stats.csv"),
         ("prof_r03_bigmaps/trace_kernel_stats.csv", "r03_n_bigmaps_kernel_stats_all_shapes.csv"),
         ("r03_fwd_pmc_after.csv", "r03_p_fwd_band_pmc_after_row_stride_fix.csv"),
         ("r03_bench_20.json", "r03_q_bench_line_steps20.json"), ("r03_bench_200.json", "r03_q_bench_line_steps200.json"),
         ("r03_bench_gpus2_gloo.json", "r03_q_bench_line_gpus2_self_launched_gloo.json"),
         ("r03_ab_round2_vs_round3.txt", "r03_r_ab_round2_vs_round3_kernels.txt"),
         ("r03_bigmaps_final.jsonl", "r03_e_bigmaps_tile_kernels.jsonl"), ("r03_shape_sweep.jsonl", "r03_g_shape_sweep.jsonl"),
         ("r03_fused_callers.jsonl", "r03_i_fused_callers.jsonl"), ("r03_train.jsonl", "r03_s_train_step_1gpu.jsonl"),
         ("r03_multistage.jsonl", "r03_t_multistage_train_step_tile_vs_generic.jsonl"),
         ("r03_tile_backward_stores_ab.txt", "r03_w_tile_backward_stores_ab.txt"),
         ("r03_tile_vs_table.jsonl", "r03_x_tile_vs_table_kernels.jsonl"),
         ("r03_tile_pmc_16x112.csv", "r03_y_tile_kernels_pmc_after_rewrite.csv")]
for a, b in pairs:
    shutil.copy(os.path.join("gpurun_out", a), os.path.join("profiles", b))
out = subprocess.run([sys.executable, "scripts/summarize_bigmaps_trace.py", "gpurun_out/prof_r03_bigmaps/trace_kernel_trace.csv"],
                     capture_output=True, text=True).stdout
open("profiles/r03_j_bigmaps_rocprof_per_shape.csv", "w").write(out)
print(out)
