#!/bin/bash
# One GPU-box visit (round 6 on: ONE parameterised script instead of a script per visit -- tools/sessions/ is the archive).
#   usage: tools/visit.sh <tag> [step ...]
#   steps: test            the whole `-m gpu` suite                      testk:<expr>   pytest -k <expr> (-m gpu)
#          bench           the driver's command (bench.py --steps 20 --warmup 5): compact line + full record
#          trace:<name>:<bench args with , for space>    rocprofv3 --kernel-trace --stats -> <name>_kernel_trace_summary.txt
#          pmc:<name>:<bench args>                       tools/pmc.sh passes -> pmc_<name>.txt
#          py:<script with , for space>                  any tool under tools/ (python), stdout -> <first word>.txt
#          env:<NAME=VALUE>                              export for the steps that follow (e.g. env:MPE_HIP_LIB=...)
#          rtrace:<name>:<python script with , for space>   rocprofv3 --kernel-trace --stats of any tool -> <name>_kernel_trace_summary.txt
#          rpmc:<name>:<kernel-name pattern, _ for space>:<python script>   FETCH_SIZE / WRITE_SIZE passes of any tool -> pmc_<name>.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-v}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp PYTHONPATH=$R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    test)  ( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1 ) 2> $O/pytest.time
           echo "pytest rc=$?"; tail -14 $O/pytest.log | cut -c1-250; grep real $O/pytest.time ;;
    testk) ( time timeout 1500 python -m pytest tests -m gpu -x -q -k "$rest" > $O/pytest_k.log 2>&1 ) 2> $O/pytest_k.time
           echo "pytest -k rc=$?"; tail -12 $O/pytest_k.log | cut -c1-250 ;;
    bench) ( time timeout 900 python bench.py --steps 20 --warmup 5 --full-json $O/bench_full.json > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench.time
           echo "bench rc=$? line bytes: $(wc -c < $O/bench_line.json)"; cat $O/bench_line.json; grep real $O/bench.time ;;
    trace) name=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""; a=${a//,/ }
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o x -- \
              python $R/bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 --steps 200 $a > $O/trace_$name.line 2> $O/trace_$name.err)
           f=$(find $O/trace_$name -name "x_kernel_trace.csv" | head -1)
           python tools/trace_summary.py $f "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra --repeats 2 --region-ms 40 --steps 200 $a" > $O/${name}_kernel_trace_summary.txt
           cp $(find $O/trace_$name -name "x_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
           rm -rf $O/trace_$name; head -8 $O/${name}_kernel_trace_summary.txt | cut -c1-200; tail -3 $O/${name}_kernel_trace_summary.txt ;;
    pmc)   name=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""; a=${a//,/ }
           PMC_TRAFFIC_ONLY=${PMC_TRAFFIC_ONLY:-1} tools/pmc.sh ${TAG}_$name $a > $O/pmc_$name.log 2>&1; tail -3 $O/pmc_$name.log ;;
    py)    cmd=${rest//,/ }; w=${cmd%% *}; w=$(basename $w .py)
           timeout 1200 python $cmd > $O/$w.txt 2> $O/$w.err; echo "$cmd rc=$?"; tail -40 $O/$w.txt | cut -c1-220; tail -5 $O/$w.err | cut -c1-300 ;;
    env)   export "$rest"; echo "export $rest" ;;
    rtrace) name=${rest%%:*}; cmd=${rest#*:}; cmd=${cmd//,/ }
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o x -- python $R/$cmd > $O/trace_$name.log 2>&1)
           f=$(find $O/trace_$name -name "x_kernel_trace.csv" | head -1)
           python tools/trace_summary.py $f "rocprofv3 --kernel-trace --stats -- python $cmd" > $O/${name}_kernel_trace_summary.txt
           cp $(find $O/trace_$name -name "x_kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
           rm -rf $O/trace_$name; head -8 $O/${name}_kernel_trace_summary.txt | cut -c1-200 ;;
    rpmc)  name=${rest%%:*}; r2=${rest#*:}; pat=${r2%%:*}; pat=${pat//_/ }; cmd=${r2#*:}; cmd=${cmd//,/ }; mkdir -p $O/pmc_$name
           for grp in FETCH_SIZE WRITE_SIZE; do
             (cd /tmp && timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$name/$grp -o x -- python $R/$cmd > $O/pmc_$name/$grp.log 2>&1)
           done
           python profiles/pmc_summary.py $O/pmc_$name "$pat" > $O/pmc_$name.txt; rm -f $O/pmc_$name/*/x_kernel_trace.csv; cat $O/pmc_$name.txt | cut -c1-200 ;;
    *)     echo "unknown step $step" ;;
  esac
done
ls $O
