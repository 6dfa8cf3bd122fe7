mkdir -p gpurun_out/t13
timeout 500 python profiles/microbench/probes/probe_leak.py > gpurun_out/t13/leak.txt 2>&1
grep -v "amdgpu.ids\|^W10" gpurun_out/t13/leak.txt | cut -c1-300 | tail -20
