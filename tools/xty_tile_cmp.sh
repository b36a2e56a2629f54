# f32 Gram kernel, 128-wide (NIF_XTY_FRAGS=2) against 192-wide (=3) tiles, and the f64 matrix path, at the shapes the
# end-to-end workloads run (C3, C5, the normal equations of C1)
for spec in "4096 8192" "25000 6144 --sym" "200000 6144 --sym" "25000 24576 --sym" "100000 1024" "20000 2048"; do
  set -- $spec
  for f in 2 3; do
    echo -n "frags=$f: "; NIF_XTY_FRAGS=$f timeout -k 10 100 python tools/xty_bench.py --m $1 --n $2 $3 2>&1 | grep "^xty"
  done
done
for spec in "6400 4609 --sym" "25600 2305 --sym" "102400 577 --sym" "1600 4609"; do
  set -- $spec
  timeout -k 10 100 python tools/xty_bench.py --m $1 --n $2 $3 --acc f64 2>&1 | grep "^xty"
done
