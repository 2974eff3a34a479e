cd $GRAFT_REPO_ROOT
out=gpurun_out/r06u; mkdir -p $out
( echo "scripts/fuzz_campaign.py on the final library of round 6 (quiet rows of the relocate scan; counters before guide values in the reload after a move):"
  echo "  --cases 12000 --seed 71 --max_k 40 --min_n 8 --max_n 33:"; timeout 900 python scripts/fuzz_campaign.py --cases 12000 --seed 71 --max_k 40 --min_n 8 --max_n 33 2>&1 | tail -1
  echo "  --cases 6000 --seed 72 --max_k 30 --max_n 130:"; timeout 900 python scripts/fuzz_campaign.py --cases 6000 --seed 72 --max_k 30 --max_n 130 2>&1 | tail -1
  echo "  --cases 1500 --seed 73 --max_k 12 --min_n 128 --max_n 255:"; timeout 900 python scripts/fuzz_campaign.py --cases 1500 --seed 73 --max_k 12 --min_n 128 --max_n 255 2>&1 | tail -1
  echo "  --cases 1500 --seed 74 --max_k 20 --min_n 128 --max_n 163 (quiet rows on the four-slot builds of the LDS-penalty store):"; timeout 900 python scripts/fuzz_campaign.py --cases 1500 --seed 74 --max_k 20 --min_n 128 --max_n 163 2>&1 | tail -1 ) > $out/r06_fuzz_campaign.txt
cat $out/r06_fuzz_campaign.txt
( echo "scripts/headline_parity.py --sample 32; --guide regret_pred --sample 16; --n 200 --batch 256 --time_limit 3   (final library of round 6)"
  timeout 900 python scripts/headline_parity.py --sample 32 2>&1 | tail -1
  timeout 900 python scripts/headline_parity.py --sample 16 --guide regret_pred 2>&1 | tail -1
  timeout 900 python scripts/headline_parity.py --n 200 --batch 256 --time_limit 3 2>&1 | tail -1 ) > $out/r06_headline_parity.txt
cat $out/r06_headline_parity.txt
