for v in nozero nodiv nostore; do
  export SPH_LIB_PATH=$PWD/cudafluidsimulator_amd/variants/libsph_hip_$v.so
  bash scripts/gpu_ktrace.sh r03_hist_$v 20 --warmup 5 | grep "radix_hist<10, true"
done
