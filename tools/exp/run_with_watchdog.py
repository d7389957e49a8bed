import faulthandler, runpy, sys
faulthandler.dump_traceback_later(90, exit=True)
sys.argv = ["tools/bench_nn.py", "dnn", "--frames", "65536"]
runpy.run_path("tools/bench_nn.py", run_name="__main__")
