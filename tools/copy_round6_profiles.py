#!/usr/bin/env python3
"""Copy what tools/round6_all.sh left in gpurun_out/ into profiles/ (run in the repo root after the gpurun call)."""
import os, shutil, subprocess, sys
g, p = 'gpurun_out/', 'profiles/'
cp = [('r06/pmc_traffic.json', 'pmc_traffic.json'), ('r06/bench_under_rocprof.json', 'r06_bench_under_rocprof.json'), ('r06/kernel_stats.csv', 'r06_bench_kernel_stats.csv'),
      ('r06_eemflow_plus_switches.txt', None), ('r06_eemflow_plus_timeline.txt', None), ('r06_eemflow_plus_timeline_no_wnc.txt', None), ('r06_eraft_wnc.txt', None),
      ('r06_train_switches.txt', None), ('r06_train_timeline.txt', None), ('r06_train_timeline_1280x720_b8.txt', None), ('r06_train_timeline_round5_forms.txt', None),
      ('r06_gpu_tests.txt', None), ('r06_marginal.txt', None), ('r06_rows.txt', None), ('r06_wgrad_bench.txt', None), ('r06_wgrad_nomfma.txt', None),
      ('r06_plus.txt', 'r06_eemflow_plus.txt'), ('r06_ertrain.txt', 'r06_eraft_train_step.txt'), ('r06_pipeline_per_frame.txt', None),
      ('r06_plus/eemflow_plus_kernel_stats.csv', 'r06_eemflow_plus_kernel_stats.csv'), ('r06_ertrain/eraft_train_step_kernel_stats.csv', 'r06_eraft_train_step_kernel_stats.csv'),
      ('r06_rows/train_kernel_stats.csv', 'r06_train_step_kernel_stats.csv'), ('r06_rows/voxel_kernel_stats.csv', 'r06_voxelizer_kernel_stats.csv'),
      ('r06_rows/eraft_b4_kernel_stats.csv', 'r06_eraft_b4_kernel_stats.csv')]
for a, b in cp:
    if os.path.exists(g + a): shutil.copy(g + a, p + (b or a))
    else: print('MISSING', a)
subprocess.run([sys.executable, 'tools/pmc_means.py', g + 'r06/pmc', p + 'r06_pmc_'], check=True)
