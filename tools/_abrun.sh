for p in f64 f32; do
NPX=1024 PREC=$p tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_pc0.so fast_amd/libfastmc_nopipe.so fast_amd/libfastmc.so fast_amd/libfastmc_pc0.so fast_amd/libfastmc_nopipe.so 2>&1 | grep "rows "
done
