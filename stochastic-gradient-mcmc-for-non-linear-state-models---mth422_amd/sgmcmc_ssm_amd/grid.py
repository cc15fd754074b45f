"""Whole-GPU particle-filter windows kept resident on the device (N > 16384 particles per window).

The reference's bias experiments take the mean of ten `helper.pf_gradient_estimate(pf='poyiadjis_N', N=1000000)` runs on a
buffered 48-step window as their ground truth (nonlinear_ssm_pf_experiment_scripts/gradient_error_fig_scripts/
svm_grad_compare.py:58-82).  Through the drop-in Helper that is one `pfg_run_batch` call per estimate (host buffers,
synchronous).  `ResidentWindows` is the resident form of the same computation: B windows (e.g. the ten repetitions, or a
grid of parameter vectors) with observations, parameters, results and the per-window state scratch in HBM, launched with
`pfg_launch_device_grid` on a caller stream -- T + 2 kernel launches, no host synchronisation -- with the device
generator keyed by (seed, stream0 + b, step): `launch()` again draws a fresh repetition.  bench.py times this.

PyTorch is device memory and streams only."""
import numpy as np
import torch

from . import _capi


class ResidentWindows(object):
    def __init__(self, model, observations, thetas, N, kernel=None, pf="poyiadjis_N", lambduh=None, t1=0, tL=None,
                 weights=None, prior_mean=0.0, prior_var=1.0, stat="score", dtype="f64", seed=0, stream0=0, flags=0,
                 device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("ResidentWindows needs an MI355X (no CPU fallback)")
        self.model, self.N, self.dtype = model, int(N), dtype
        self.kernel = kernel or {"svm": "prior", "garch": "optimal", "lgssm": "optimal"}[model]
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.ctx = _capi.default_context(self.device.index)
        self._launch_smoother = "nemeth"
        if pf == "poyiadjis_N":
            smoother, lam = "nemeth", 1.0
            if stat == "score":
                self._launch_smoother = "poyiadjis_n"   # launch-level statement: the score-only twin of the timestep kernel
        elif pf == "nemeth":
            smoother, lam = "nemeth", 0.95 if lambduh is None else float(lambduh)
        elif pf == "filter":
            smoother, lam = "filter", 1.0
        else:
            raise NotImplementedError("whole-GPU windows are built for pf = 'poyiadjis_N' | 'nemeth' | 'filter'")
        y = np.ascontiguousarray(observations, dtype=np.float64).reshape(-1)
        th = np.ascontiguousarray(thetas, dtype=np.float64).reshape(-1, _capi.THETA_DIM[model])
        self.T, self.B, self.P = y.shape[0], th.shape[0], th.shape[1]
        tL = self.T if tL is None else min(int(tL), self.T)
        sb = self.ctx.scratch_bytes(model, dtype, "device", self.N)
        if sb <= 0:
            raise NotImplementedError("N = {0}: whole-GPU windows serve 16384 < N <= 4194304".format(self.N))
        dev = self.device
        self.y_dev = torch.from_numpy(y).to(dev)
        th4 = np.zeros((self.B, _capi.MAX_THETA))
        th4[:, :self.P] = th
        self.theta_dev = torch.from_numpy(th4).to(dev)
        self.out_dev = torch.zeros((self.B, _capi.OUT_DOUBLES), dtype=torch.float64, device=dev)
        self.step_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.scratch_dev = torch.empty(self.B * sb, dtype=torch.uint8, device=dev)
        self.weights_dev = None
        d = np.zeros(self.B, dtype=_capi.DEV_PROBLEM_DTYPE)
        idx = np.arange(self.B, dtype=np.uint64)
        d["y"] = self.y_dev.data_ptr()
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
            if w.shape[0] < tL - int(t1):
                raise ValueError("weights shorter than tL - t1")
            self.weights_dev = torch.from_numpy(w).to(dev)
            d["weights"] = self.weights_dev.data_ptr()
        d["theta"] = self.theta_dev.data_ptr() + idx * np.uint64(8 * _capi.MAX_THETA)
        d["out"] = self.out_dev.data_ptr() + idx * np.uint64(8 * _capi.OUT_DOUBLES)
        d["scratch"] = self.scratch_dev.data_ptr() + idx * np.uint64(sb)
        d["step_ctr"] = self.step_ctr.data_ptr()
        d["prior_mean"], d["prior_var"], d["lambduh"] = float(prior_mean), float(prior_var), lam
        d["seed"] = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
        d["stream"] = np.uint64(int(stream0)) + idx
        d["T"], d["t1"], d["tL"], d["N"] = self.T, int(t1), tL, self.N
        d["smoother"], d["stat"], d["flags"] = _capi.SMOOTHER[smoother], _capi.STAT[stat], int(flags)
        self._desc = d
        self.desc_dev = torch.from_numpy(d.view(np.uint8).reshape(self.B, -1)).to(dev)
        self.launches = 0
        self._graphs = {}

    def launch(self, stream=None):
        """One repetition of every window: T + 2 launches on `stream` (default: torch's current stream), asynchronous."""
        st = stream or torch.cuda.current_stream(self.device)
        self.ctx.launch_device_grid_smoother(self.model, self.kernel, self.dtype, "device", self._launch_smoother, self.N, self.T,
                                             self.ctx.GRID_PHASE_ALL, self.B, self.desc_dev.data_ptr(), st.cuda_stream)
        with torch.cuda.stream(st):
            self.step_ctr += 1          # the next launch draws a fresh repetition (device-side key, no host sync)
        self.launches += 1

    def launch_graph(self, repetitions=1):
        """`repetitions` repetitions of every window as ONE hipGraph launch (captured once per repetition count; the
        T + 2 kernels and the counter bump of a repetition are graph nodes, every input that changes lives in HBM): the
        same computation as `repetitions` calls of launch(), bitwise, without the host issuing (T + 3) launches each.
        For callers whose host thread is busy; it buys no GPU time (measured, tools/ab/grid_graph_time.py: 0.574 / 0.572 ms
        per repetition at N = 20000, 9.475 / 9.471 at 10 x 10^6 -- the eager launches are already back to back, what
        separates them is the dependent kernel boundary on the device).  Asynchronous, on the graph's own capture stream
        (ordered after torch's current stream)."""
        g = self._graphs.get(repetitions)
        if g is None:
            if self.launches == 0:          # one eager launch first (code-object load, LDS-size attributes), undone afterwards
                ctr = self.step_ctr.clone()
                self.launch()
                torch.cuda.synchronize(self.device)
                self.step_ctr.copy_(ctr)
                self.launches = 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(repetitions):
                    self.launch()
            self.launches -= repetitions        # capturing ran nothing
            self._graphs[repetitions] = g
        g.replay()
        self.launches += repetitions

    def launch_timed(self, stream=None):
        """The same launches, one `pfg_launch_device_grid_phase` call per timestep with a pair of HIP events around each
        step kernel: returns the list of (start, end) events (read them after a synchronise).  Same numbers as launch()."""
        st = stream or torch.cuda.current_stream(self.device)
        args = (self.model, self.kernel, self.dtype, "device", self._launch_smoother, self.N, self.T)
        self.ctx.launch_device_grid_smoother(*args, self.ctx.GRID_PHASE_INIT, self.B, self.desc_dev.data_ptr(), st.cuda_stream)
        events = []
        for t in range(self.T):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            self.ctx.launch_device_grid_smoother(*args, t, self.B, self.desc_dev.data_ptr(), st.cuda_stream)
            b.record(st)
            events.append((a, b))
        self.ctx.launch_device_grid_smoother(*args, self.ctx.GRID_PHASE_FINISH, self.B, self.desc_dev.data_ptr(), st.cuda_stream)
        with torch.cuda.stream(st):
            self.step_ctr += 1
        self.launches += 1
        return events

    def results(self):
        """(mean statistic [B, h], log-likelihood [B]) of the latest launch (synchronises)."""
        o = self.out_dev.cpu().numpy()
        return o[:, :_capi.STAT_DIM[self.model]].copy(), o[:, 4].copy()
