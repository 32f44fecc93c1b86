#!/bin/bash
# the drop-in calls of the reference's inner loop with device-resident and with host (PCIe-inclusive) batches: DESIGN.md section 6
python profiles/bench_host_path.py 2>/dev/null | tail -1 > gpurun_out/host_path_r02.json; cat gpurun_out/host_path_r02.json
