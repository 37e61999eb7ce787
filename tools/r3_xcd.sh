export PYTHONPATH=.
for L in build_ab/libhens_xcd.so build_ab/libhens_nt2.so; do
for shape in "16 4096 32"; do
  set -- $shape
  for x in 0 1 0 1; do
    if [ $x = 1 ]; then export HENS_XCD=1; else unset HENS_XCD; fi
    echo -n "$L shape $shape xcd $x: "
    HENS_LIB=$L python tools/quick_bench.py --T $1 --W $2 --D $3 --steps 4000 --prof 1 | sed -n '1p;3p' | tr '\n' ' '; echo
  done
done
done
