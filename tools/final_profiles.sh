set -u
cd $GRAFT_REPO_ROOT
bash tools/run_configs.sh gpurun_out/r03f_results > gpurun_out/r03f_configs.log 2>&1
bash tools/profile_gpu.sh r03f_text_p16 --workload text > gpurun_out/r03f_prof_text.log 2>&1
bash tools/profile_gpu.sh r03f_low_p16 --workload low > gpurun_out/r03f_prof_low.log 2>&1
bash tools/profile_gpu.sh r03f_urls_p16 --workload urls > gpurun_out/r03f_prof_urls.log 2>&1
bash tools/profile_gpu.sh r03f_page_p13 --workload page > gpurun_out/r03f_prof_page.log 2>&1
bash tools/profile_gpu.sh r03f_text_p15 --workload text --p 15 > gpurun_out/r03f_prof_text15.log 2>&1
bash tools/pmc_insts.sh r03f_text_p16 --workload text > gpurun_out/r03f_insts_text.log 2>&1
bash tools/pmc_insts.sh r03f_low_p16 --workload low > gpurun_out/r03f_insts_low.log 2>&1
cat gpurun_out/r03f_configs.log
