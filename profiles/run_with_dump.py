"""Run a script with faulthandler bound to SIGUSR1 (`timeout -s USR1 …` then prints where every thread stands).
   python profiles/run_with_dump.py bench.py --flags…"""
import faulthandler, runpy, signal, sys
faulthandler.register(signal.SIGUSR1, all_threads=True)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
