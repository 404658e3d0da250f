for cfg in "0 0 0" "0 0 2" "8 0 0" "8 0 1" "0 200 0" "0 200 1" "8 200 0"; do
  t0=$(date +%s%N); tools/exitcost $cfg; t1=$(date +%s%N); echo "  [$cfg] wall $(( (t1 - t0) / 1000000 )) ms"
done
