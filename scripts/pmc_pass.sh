# usage: bash scripts/pmc_pass.sh <tag> <counter> [<counter> ...]   -- one rocprofv3 --pmc pass over scripts/pmc_dict.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 90 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/pmc_dict.py 1024 > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python scripts/pmc_parse.py gpurun_out/pmc_$tag ALL | awk '{k=$0; sub(/^ *[0-9]+ /,"",k); print k}' | sort | uniq | tail -40
