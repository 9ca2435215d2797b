for p in f64 f32; do
NPX=1024 PREC=$p tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_c5.so fast_amd/libfastmc_c4.so fast_amd/libfastmc_c6.so fast_amd/libfastmc.so fast_amd/libfastmc_c5.so 2>&1 | grep "rows "
done
