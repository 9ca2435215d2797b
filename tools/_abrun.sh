for s in 128 256 512; do
NPX=$s PREC=f64 tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_rwpe.so fast_amd/libfastmc.so fast_amd/libfastmc_rwpe.so 2>&1 | grep "rows "
done
