#!/bin/bash
# Common prologue of a GPU visit: some boxes of the pool hand out a GPU on which the FIRST device allocation of every process
# aborts ("Memory access fault by GPU node-2", sessions 4 and 7 of round 4).  Detect that in a few seconds and leave, instead of
# burning the visit's budget on timeouts.
timeout 120 python - <<'PY'
import torch
x = torch.zeros(1 << 20, device="cuda")
y = (x + 1).sum().item()
assert y == float(1 << 20), y
print("gpu ok:", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
PY
