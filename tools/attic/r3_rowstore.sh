export PYTHONPATH=.
for rep in 1 2; do
for L in rs1 rs0 rs2; do
  echo -n "$L: "
  HENS_LIB=build_ab/libhens_$L.so python tools/quick_bench.py --T 16 --W 4096 --D 32 --steps 4000 --prof 1 | sed -n '1p;2p' | tr '\n' ' ' | sed 's/T=16 W=4096 D=32: 4000 iters in//' | sed "s/'pt_ms.*n_iters': 4000,//"; echo
done; done
