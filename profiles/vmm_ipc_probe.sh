#!/bin/bash
# exporter (standalone, system runtime) + importer: first the C++ one, then the PyTorch one.  $1 = chunks, $2 = chunk bytes
hipcc -O2 --offload-arch=gfx950 profiles/vmm_ipc_probe.cpp -o /tmp/vmm_probe || exit 1
for imp in cpp torch; do
  sock=/tmp/vmm_probe_$imp.sock; rm -f $sock
  timeout -k 5 90 /tmp/vmm_probe export $1 $2 $sock & ep=$!
  if [ $imp = cpp ]; then timeout -k 5 60 /tmp/vmm_probe import $1 $2 $sock; else timeout -k 5 80 python3 profiles/vmm_ipc_probe_torch.py $1 $2 $sock; fi
  echo "$imp importer rc=$?"; wait $ep; echo "exporter rc=$?"
done
