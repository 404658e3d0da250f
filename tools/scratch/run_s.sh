mkdir -p gpurun_out
{
echo "nproc: $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
taskset -p $$ 
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" 
cat /sys/devices/system/node/node*/cpulist 2>/dev/null
for i in 1 2 3 4 5 6; do ./tools/packbench 2>&1 | grep -E "^ ?(8|14|16) threads" | head -3 | tr '\n' ' '; echo; done
} > gpurun_out/s_probe.log 2>&1
cat gpurun_out/s_probe.log
