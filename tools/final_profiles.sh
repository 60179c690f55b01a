set -u
cd $GRAFT_REPO_ROOT
bash tools/run_configs.sh gpurun_out/r03g_results > gpurun_out/r03g_configs.log 2>&1
bash tools/profile_gpu.sh r03g_text_p16 --workload text > gpurun_out/r03g_prof_text.log 2>&1
bash tools/profile_gpu.sh r03g_low_p16 --workload low > gpurun_out/r03g_prof_low.log 2>&1
bash tools/profile_gpu.sh r03g_urls_p16 --workload urls > gpurun_out/r03g_prof_urls.log 2>&1
bash tools/profile_gpu.sh r03g_page_p13 --workload page > gpurun_out/r03g_prof_page.log 2>&1
bash tools/profile_gpu.sh r03g_text_p15 --workload text --p 15 > gpurun_out/r03g_prof_text15.log 2>&1
bash tools/pmc_insts.sh r03g_text_p16 --workload text > gpurun_out/r03g_insts_text.log 2>&1
bash tools/pmc_insts.sh r03g_low_p16 --workload low > gpurun_out/r03g_insts_low.log 2>&1
cat gpurun_out/r03g_configs.log
