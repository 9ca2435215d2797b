for s in 1024; do for p in f64 f32; do
NPX=$s PREC=$p tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_nopk.so fast_amd/libfastmc_nobatch.so fast_amd/libfastmc.so fast_amd/libfastmc_nopk.so 2>&1 | grep "rows "
done; done
for s in 256 512; do NPX=$s PREC=f64 tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_nobatch.so fast_amd/libfastmc.so fast_amd/libfastmc_nobatch.so 2>&1 | grep "rows "; done
