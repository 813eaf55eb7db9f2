# In-kernel last-arrival merge of the split-context attention against the second (merge) launch, and the split threshold.
B="python bench.py --no-cpu-baseline --no-extra --no-kernel-leg"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "tok/s")'
for ctx in 1024 2048 8192; do
  echo "context $ctx merge-launch: $(GBX_ATTN_MERGE_LAUNCH=1 $B --context $ctx --steps 128 2>/dev/null | python -c "$P")   in-kernel: $($B --context $ctx --steps 128 2>/dev/null | python -c "$P")"
done
for min in 1024 512 256; do
  echo "default bench (context 128 + 512 steps), split from capacity $min: $(GBX_ATTN_SPLIT_MIN=$min $B 2>/dev/null | python -c "$P")"
  echo "context 384 + 64 steps, split from capacity $min: $(GBX_ATTN_SPLIT_MIN=$min $B --context 384 --steps 64 2>/dev/null | python -c "$P")"
done
