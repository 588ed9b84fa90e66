for spec in "5000 1000 5000" "5000 1000 1000" "5000 500 1000" "10000 500 500" "10000 1000 500" "10000 1000 1000" "10000 5000 1000"; do
  for f in 0 1 2 4 24 21; do ./tools/micro/pp_shape_bench $spec $f 1; done
done
