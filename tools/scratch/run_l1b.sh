for cs in 32 16 8; do echo "== cs $cs"; PCR_L1B_CS=$cs python tools/train_detail.py 2>/dev/null | grep -E "sa_l1_bwd" | cut -c1-110; done
