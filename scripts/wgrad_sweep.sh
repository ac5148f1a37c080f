for b in 256 384 512 768 1024; do for c in 12 16 24 32 48; do
  echo -n "blocks=$b minchunks=$c: "; FRX_WGRAD_BLOCKS=$b FRX_WGRAD_MINCHUNKS=$c python scripts/layer_times.py 256 2>/dev/null | tail -1
done; done
