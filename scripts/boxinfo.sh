echo "host: $(hostname)"; python - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print("gpu:", p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory/2**30,1), "uuid", getattr(p,'uuid',None))
PY
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4
