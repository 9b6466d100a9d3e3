cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05f
timeout -k 10 200 python3 scripts/ei_pass_timing.py 20 24 | tail -4
rm -rf gpurun_out/r05f/pmc_ei
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r05f/pmc_ei -- python3 scripts/ei_pass_timing.py 24 > gpurun_out/r05f/pmc_ei.log 2>&1
python3 scripts/pmc_kernel_means.py gpurun_out/r05f/pmc_ei acq_kernel
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "acquisition or sweep or ragged or edge or argmax or trial or sets or nan or golden or oracle or toy or smoke" 2>&1 | tail -3
