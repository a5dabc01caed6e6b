"""How many compute units this process's workgroups land on (riab_probe_compute_units), as the runtime reports it and as
counted — run under HSA_CU_MASK / ROC_GLOBAL_CU_MASK to see what a masked process gets:
    HSA_CU_MASK=0:0-127 python tools/cu_mask_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ratinabox_amd as riab  # noqa: E402

L = riab._lib
props = torch.cuda.get_device_properties(0)
scratch = torch.zeros(L.CU_PROBE_WORDS, dtype=torch.int32, device="cuda")
L.check(L.lib.riab_probe_compute_units(L.ptr(scratch), L.current_stream()), "probe")
torch.cuda.synchronize()
w = scratch.cpu().numpy()
per_xcc = [int((w[256 * x:256 * x + 256] != 0).sum()) for x in range(16)]
print({"HSA_CU_MASK": os.environ.get("HSA_CU_MASK"), "ROC_GLOBAL_CU_MASK": os.environ.get("ROC_GLOBAL_CU_MASK"),
       "multi_processor_count": props.multi_processor_count, "counted": int(w[-1]), "per_xcc": per_xcc})
