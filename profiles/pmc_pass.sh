#!/bin/bash
# One rocprofv3 counter pass with a watchdog:  pmc_pass.sh SECONDS LOGFILE rocprofv3 ARGS...
# rocprofv3 REFUSES a counter set that does not fit one pass ("... exceeds the capabilities of the hardware ...": the L2
# takes four counters a pass on gfx950, the SQ eight) -- and the profiled process then does not end by itself: in round 5 such a
# pass sat until its 200-second timeout and cost a GPU call a quarter of an hour.  This wrapper starts the pass
# as its own child, reads the log once a second, and ends exactly that child (its PID, no pattern) the moment the refusal
# shows or the time is up.  Exit status: the pass's, 124 for a timeout, 125 for a refused counter set.
LIMIT=$1; LOG=$2; shift 2
setsid "$@" > "$LOG" 2>&1 &   # its own process group: the pass and whatever it started end together (kill -- -PGID, nothing by pattern)
PID=$!
for ((t = 0; t < LIMIT; t++)); do
    if ! kill -0 $PID 2>/dev/null; then wait $PID; exit $?; fi
    if grep -q -i "exceeds the capabilities\|cannot be collected in single pass\|Unable to find counter" "$LOG" 2>/dev/null; then
        kill -- -$PID 2>/dev/null; sleep 2; kill -9 -- -$PID 2>/dev/null; wait $PID 2>/dev/null
        echo "pmc_pass: counter set refused: $(grep -m1 -i 'exceeds the capabilities\|single pass\|Unable to find counter' "$LOG" | cut -c1-200)" >&2
        exit 125
    fi
    sleep 1
done
kill -- -$PID 2>/dev/null; sleep 5; kill -9 -- -$PID 2>/dev/null; wait $PID 2>/dev/null
echo "pmc_pass: no end after $LIMIT s" >&2
exit 124
