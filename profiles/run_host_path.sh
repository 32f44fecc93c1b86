#!/bin/bash
# the reference's literal inner loop against the fused step, and the PCIe-inclusive host-fed rate: DESIGN.md section 6
python profiles/bench_host_path.py 2>gpurun_out/host_path_r06.err | tail -1 > gpurun_out/host_path_r06.json; cat gpurun_out/host_path_r06.json
