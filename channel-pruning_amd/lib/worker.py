"""Process-isolation helper with the reference's interface (lib/worker.py:4-68).

``Worker().do(target, **kwargs)`` forks a child, runs ``target(**kwargs)`` there, requires a
dict back and returns it.  The GPU context is created inside the child (after the fork), as
the reference does for its Caffe context (lib/net.py:55-58).  Differences, all on the error
path: a child that dies or raises no longer deadlocks the parent on ``queue.get()``
(worker.py:64) -- the exception text is forwarded and re-raised; ``device=`` pins the child
to one GPU (the per-GPU launcher of cpmi355.shard uses it)."""
import multiprocessing as mp
import os
import traceback


class Worker():
    def __init__(self, device=None):
        self._ctx = mp.get_context("fork")   # closures are valid targets, as in the reference
        self.mp_queue = self._ctx.Queue()
        self._target = None
        self.device = device

    def set_target(self, target):
        self._target = target

    def do(self, target=None, **mp_kwargs):
        if target is None:
            assert self._target is not None, "please provide target"
            target = self._target
        else:
            self._target = target
        device = self.device

        def job(queue, **kwargs):
            try:
                if device is not None:
                    os.environ["CP_DEVICE"] = str(device)
                ret = target(**kwargs)
                assert isinstance(ret, dict)
                queue.put(("ok", ret))
            except BaseException as e:  # forward instead of hanging the parent
                queue.put(("err", "%s\n%s" % (repr(e), traceback.format_exc())))

        mp_kwargs['queue'] = self.mp_queue
        p = self._ctx.Process(target=job, kwargs=mp_kwargs)
        p.start()
        while True:
            try:
                status, payload = self.mp_queue.get(timeout=1.0)
                break
            except Exception:
                if not p.is_alive():
                    p.join()
                    raise RuntimeError("Worker child exited with code %s without a result" % p.exitcode)
        p.join()
        if status != "ok":
            raise RuntimeError("Worker target failed in child:\n" + payload)
        return payload
