# The EI pass's issue counters in two passes (2^24 candidates).  usage: bash scripts/probes/ei_pmc_detail.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/eipmc
rm -rf gpurun_out/eipmc/a gpurun_out/eipmc/b gpurun_out/eipmc/c
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/eipmc/a -- python3 scripts/ei_pass_timing.py 24 > gpurun_out/eipmc/a.log 2>&1
python3 scripts/pmc_kernel_means.py gpurun_out/eipmc/a acq_kernel
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/eipmc/b -- python3 scripts/ei_pass_timing.py 24 > gpurun_out/eipmc/b.log 2>&1
python3 scripts/pmc_kernel_means.py gpurun_out/eipmc/b acq_kernel
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/eipmc/c -- python3 scripts/ei_pass_timing.py 24 > gpurun_out/eipmc/c.log 2>&1
python3 scripts/pmc_kernel_means.py gpurun_out/eipmc/c acq_kernel
tail -3 gpurun_out/eipmc/c.log
