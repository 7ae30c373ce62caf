# the nine-emulator chain at the walker counts analyses use with emcee (examples/RunBayesianAnalysis.ipynb: nwalkers = 100) up to 4096
R=$GRAFT_REPO_ROOT
for w in 100 256 512 1024 2048; do python3 $R/tools/gpu_multi_chain_profile.py 200 --walkers=$w; done
