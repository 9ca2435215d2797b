for p in f64 f32; do
NPX=1024 PREC=$p tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_w16.so fast_amd/libfastmc_w16c0.so fast_amd/libfastmc_w16nb.so fast_amd/libfastmc_twg.so fast_amd/libfastmc.so 2>&1 | grep "rows "
done
