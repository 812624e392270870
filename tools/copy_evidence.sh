#!/bin/bash
# copy the judged files of a tools/evidence_r4.sh run (gpurun_out/evidence_<tag>/) into profiles/:  tools/copy_evidence.sh <tag> [round dir, default r4]
E=gpurun_out/evidence_$1; R=profiles/${2:-r4}; mkdir -p $R
for f in bench_default_command.json bench_time.txt bf16_block_level0_kernels.txt bf16_block_level3_kernels.txt dcpt_allbf16_128_kernels.txt dcpt_allbf16_256_kernels.txt ddp_probe.txt kernel_stats_serialized.csv level_probe_bf16.txt level_probe_fp32.txt naf_bf16_kernels.txt naf_bf16_kernels_serialized.txt pmc_tn256_level3.txt pytest_gpu_full.log restormer_balanced_kernels.txt restormer_auto_kernels.txt pmc_chain_level3.txt fp32_power_ceiling.txt fuzz_seed6_tail.txt rocprofv3_summary_serialized.txt step_budget_default_command.txt tn256_probe.txt x3_step_kernels.txt infer2k_bf16_kernels_two_streams.txt infer2k_streams.txt stream_stress.txt fuzz_seed3_tail.txt; do cp $E/$f $R/ 2>/dev/null || echo missing $f; done
cp $E/extra_*.json $R/; cp $E/step_budget.json $R/step_budget_default_command.json; cp $E/step_budget.json profiles/step_budget.json; cp $E/pmc_traffic.json profiles/pmc_traffic.json
python tools/showbench.py $R/bench_default_command.json | head -3
