# source me on the GPU box:  gstep <seconds> <log> <command...>
# One GPU step under its own timeout; its output goes to <log>.  An ordinary failure (a failing test, a Python error) is reported and the
# chain goes on; a step that had to be KILLED (timeout: 124 / 137) ends the chain -- after a hung GPU step no further GPU step is started.
gstep() {
  local t=$1 log=$2; shift 2
  echo "[gstep] $(date +%T) $*" >&2
  timeout -k 10 "$t" "$@" > "$log" 2>&1
  local rc=$?
  echo "[gstep] rc=$rc  $log" >&2
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gstep] KILLED at its limit: stopping the chain" >&2; return 1; fi
  return 0
}
