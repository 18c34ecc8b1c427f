"""Shared by the rocpd_* scripts: where in a trace of `bench.py` the PIPELINED evaluator ran.

bench.py runs, in this order: warm-up, the timed pipelined steps, sequential steps (step-time spread), the parity legs.  The END of
a trace is therefore sequential work; the pipelined steps are the kernels launched by the evaluator's slot threads (the threads other
than the main one that launch whole feature passes).  Their first launches are the slots' graph captures, one after the other, so the
window is cut by launch-count percentiles: from the 35th to the 93rd percentile of the slot threads' launches lies inside the dense,
pipelined region and leaves out its drain."""


def pipelined_window(cur, lo=0.35, hi=0.93):
    per = list(cur.execute("select tid, min(start), sum(name like '%conv_igemm%') from kernels group by tid"))
    heavy = [(t, s) for t, s, n in per if n and n >= 100]
    if len(heavy) < 2:
        return None
    main = min(heavy, key=lambda v: v[1])[0]
    q = ','.join(str(t) for t, _ in heavy if t != main)
    starts = [r[0] for r in cur.execute(f"select start from kernels where tid in ({q}) order by start")]
    return starts[int(lo * len(starts))], starts[int(hi * len(starts))]
