for w in 4 3 2; do
  VTI_MASK_WGS_PER_CU=$w rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w${w}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-line --parity-frames 0 --preheat 0.2 > gpurun_out/w${w}.json 2>gpurun_out/w${w}.err
  echo "wgs/cu $w: $(grep -i masks_group gpurun_out/w${w}_prof/*/*kernel_stats.csv | cut -d, -f2-4)"
done
