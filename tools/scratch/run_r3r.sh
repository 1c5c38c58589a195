for st in 0 8 16 32 64 128; do
  echo "== stagger $st"
  PCR_STAGGER=$st PCR_STREAM_MIN=0 python tools/bench_tdense_fwd.py 32 4096 512
  PCR_STAGGER=$st PCR_STREAM_MIN=0 python tools/bench_tdense_fwd.py 64 3072 512
  PCR_STAGGER=$st PCR_STREAM_MIN=0 python tools/bench_tdense.py 32 4096 512
done
PCR_STREAM_MIN=0 python tools/bench_tdense_fwd.py 128 1536 512
