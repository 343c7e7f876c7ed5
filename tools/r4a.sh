ulimit -c 0
mkdir -p gpurun_out
T=${TAG:-r4a}
timeout -k 10 300 python3 tools/compare_envs.py "JUR_EGA_GROUP=0" "${CMP:-JUR_EGA_BLOCK=512}" 100000 > gpurun_out/${T}_cmp.log 2>&1 || { tail -20 gpurun_out/${T}_cmp.log; exit 1; }
cat gpurun_out/${T}_cmp.log
timeout -k 10 600 bash tools/ab_env.sh "$@" > gpurun_out/${T}_ab.log 2>&1
cat gpurun_out/${T}_ab.log
