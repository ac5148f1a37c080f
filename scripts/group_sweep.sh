for c in 32 48 64 96 128; do for b in 512 768; do
  echo -n "chunks=$c blocks=$b: "; FRX_WGRAD_GROUP_CHUNKS=$c FRX_WGRAD_GROUP_BLOCKS=$b python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c60-140
done; done
