cd $GRAFT_REPO_ROOT
SUHMO_DIST_BACKEND=gloo SUHMO_TRANSPORT=ipc-probe timeout -k 10 300 python3 bench.py --gpus 2 --cells 2048 --steps 5 --warmup 2 --no-cpu > gpurun_out/r04_t_bench2_ipc.json 2> gpurun_out/r04_t_bench2_ipc.err; echo "rc $?"; tail -c 1500 gpurun_out/r04_t_bench2_ipc.json; tail -5 gpurun_out/r04_t_bench2_ipc.err
SUHMO_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --cells 2048 --steps 5 --warmup 2 --no-cpu > gpurun_out/r04_t_bench2_gloo.json 2> gpurun_out/r04_t_bench2_gloo.err; echo "rc $?"; tail -c 600 gpurun_out/r04_t_bench2_gloo.json
