"""Run a script with a watchdog that dumps every thread's Python stack if it is still running after N seconds
(diagnosing the intermittent stall with TunableOp enabled):  python tools/debug/stall_trace.py 90 tools/bench_swin.py --tuned-gemms"""
import faulthandler
import runpy
import sys

secs, script = int(sys.argv[1]), sys.argv[2]
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
