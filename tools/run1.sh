WLS="driver" bash tools/profile_round.sh r06 $1 > gpurun_out/prof_driver.log 2>&1; tail -5 gpurun_out/prof_driver.log
head -12 gpurun_out/profiles_r06/r06_driver_command_kernel_stats.csv | cut -c1-200
