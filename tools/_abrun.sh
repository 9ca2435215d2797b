for s in 256 512 1024; do
NPX=$s PREC=f64 tools/abl.sh fast_amd/libfastmc.so fast_amd/libfastmc_cpw2.so fast_amd/libfastmc_cpw4.so fast_amd/libfastmc_cpw4a.so fast_amd/libfastmc.so 2>&1 | grep "rows "
done
