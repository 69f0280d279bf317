#!/bin/bash
# An observed process and a disturbing process on one GPU.
#   bash tools/share_probe.sh <observed> <disturbing> [count]
#     observed:   fdgs  = tools/fwd_stale_probe.py main (this library's forward buffers, bit by bit)
#                 torch = tools/share_probe_torch.py main (plain PyTorch operations, bit by bit)
#     disturbing: fdgs | torch = the same loops without the comparison;
#                 restart = tests/dp_gpu_worker.py started again and again; none;
#                 micro:<C> = neighbour kernel C of tools/microbench/coresident.hip in a loop
#     observed micro = the synthetic victims of tools/microbench/coresident.hip for [count] seconds
#   DIS_ENV="NAME=value ..." / OBS_ENV="..." are added to the disturbing / the observed process's environment only.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OBS=${1:-fdgs}; DIS=${2:-fdgs}; COUNT=${3:-400}
T=${PROBE_TIMEOUT:-300}              # seconds either process may take
export MASTER_ADDR=127.0.0.1 FDGS_TEST_VIEWS=${FDGS_TEST_VIEWS:-4}
D=$(mktemp -d)
case "$DIS" in
  fdgs)  env $DIS_ENV timeout -k 10 $T python -u tools/fwd_stale_probe.py load $D > $D/load.log 2>&1 & BG=$! ;;
  torch) env $DIS_ENV timeout -k 10 $T python -u tools/share_probe_torch.py load $D > $D/load.log 2>&1 & BG=$! ;;
  micro:*) env $DIS_ENV timeout -k 10 $T ./tools/microbench/coresident neighbour ${DIS#micro:} $D > $D/load.log 2>&1 & BG=$! ;;
  restart)
    ( while [ ! -e $D/ready ]; do sleep 0.2; done
      while [ ! -e $D/done ]; do
        E=$(mktemp -d); FDGS_TEST_TAG=_x timeout -k 10 100 python tests/dp_gpu_worker.py $E > /dev/null 2>&1; rm -rf $E
      done ) & BG=$! ;;
  *) BG="" ;;
esac
if [ "$OBS" = micro ]; then
  env $OBS_ENV timeout -k 10 $T ./tools/microbench/coresident victims $D $COUNT
elif [ "$OBS" = torch ]; then
  env $OBS_ENV timeout -k 10 $T python -u tools/share_probe_torch.py main $D $COUNT
else
  env $OBS_ENV timeout -k 10 $T python -u tools/fwd_stale_probe.py main $D $COUNT
fi
RC=$?
touch $D/done
[ -n "$BG" ] && wait $BG
[ -e $D/load.log ] && tail -2 $D/load.log
exit $RC
