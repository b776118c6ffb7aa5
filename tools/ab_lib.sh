for r in 1 2; do
for lib in prev new; do
  if [ $lib = prev ]; then export VPX_LIB=build/libvpx_prev.so; else unset VPX_LIB; fi
  for b in 4 32; do echo "$lib B=$b $(python bench.py --batch $b --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c60-90)"; done
done
done
