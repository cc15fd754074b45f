"""ChainEnsemble: many independent SGLD chains resident on one MI355X.

The reference runs one chain in one Python thread; its experiment grid fans chains / settings
out over processes (driver_utils.py:69-111).  On an MI355X one chain keeps one workgroup (one
of 256 CUs) busy, so the native unit of work is an *ensemble*: C chains advance in lock-step,
each SGLD step = ONE particle-filter launch (one workgroup per chain x window) + ONE tiny
update kernel (prior gradient, 1/T scaling, Langevin noise, projection).  Everything --
observations, parameters, descriptors, RNG counters, results -- stays in HBM between steps;
the host only enqueues launches.  PyTorch is used for device memory, streams and
`torch.distributed` (RCCL) only.

Multi-GPU: chains are independent, so rank r owns chains [r*C, (r+1)*C) (weak scaling) and
the only collective is the gather of the parameter samples (`gather_samples`).

Per-step semantics follow `SGMCMCSampler.sample_sgld` + `project_parameters`
(sgmcmc_sampler.py:549-567, 650-656) with `noisy_gradient(kind='pf')` (:427-464); the RNG is
the device Philox generator, so trajectories are statistically, not bitwise, equivalent to
the reference (bitwise parity is what the REPLAY mode of the Sampler classes is for).
"""
import ctypes

import numpy as np
import torch

from . import _capi
from .sgmcmc_sampler import random_subsequence_and_weights

_MODELS = {}


def _model_info(model):
    if not _MODELS:
        from .models import svm, garch, lgssm
        _MODELS.update(svm=(svm.SVMParameters, svm.SVMPrior, svm.SVMHelper),
                       garch=(garch.GARCHParameters, garch.GARCHPrior, garch.GARCHHelper),
                       lgssm=(lgssm.LGSSMParameters, lgssm.LGSSMPrior, lgssm.LGSSMHelper))
    return _MODELS[model]


def prior_hyper(model, prior):
    """Flatten a (1-D) Prior's hyper-parameters into the pfg_prior_hyper struct."""
    h = _capi.PriorHyper()
    hp = prior.hyperparams
    first = lambda v: float(np.asarray(v).reshape(-1)[0])
    h.df_Rinv, h.scale_Rinv = first(hp['df_Rinv']), first(hp['scale_Rinv'])
    h.df_Qinv = h.scale_Qinv = h.var_col_A = h.var_col_C = 1.0
    if model in ("svm", "lgssm"):
        h.df_Qinv, h.scale_Qinv = first(hp['df_Qinv']), first(hp['scale_Qinv'])
        h.mean_A, h.var_col_A = first(hp['mean_A']), first(hp['var_col_A'])
    if model == "lgssm":
        h.mean_C, h.var_col_C = first(hp['mean_C']), first(hp['var_col_C'])
    if model == "garch":
        for k in ('scale_mu', 'shape_mu', 'alpha_phi', 'beta_phi', 'alpha_lambduh', 'beta_lambduh'):
            setattr(h, k, first(hp[k]))
    return h


class ChainEnsemble(object):
    """C independent SGLD chains of one model on one series, resident on `device`.

    Args:
      model: 'svm' | 'garch' | 'lgssm';  observations: (T,) or (T,1), or a LIST of such arrays =
             independent sequences (the Seq*Sampler setting, e.g. the gap-split EURUS segments):
             every chain and step draws ONE sequence uniformly (num_sequences = 1) and a buffered
             window inside it, its gradient rescaled by T_total / T_sequence
             (sgmcmc_sampler.py:1249-1283)
      parameters: a Parameters object (all chains start there) or an array [C, P] of raw thetas
      num_chains: C (ignored when `parameters` is an array)
      N, pf ('poyiadjis_N' | 'nemeth'), lambduh, kernel: particle-filter settings
      epsilon: SGLD step size;  prior: Prior (default: the model's default prior, var=100 / 1)
      subsequence_length S / buffer_length B: -1 = full sequence (no window sampling)
      dtype: 'f64' | 'f32' particle-state arithmetic;  seed: Philox key
      chain_offset: global index of this rank's first chain (keeps streams distinct across GPUs)
      resampling: 'multinomial' (the reference's) | 'systematic' (extension, parity-unpinned)
      sampler: 'sgld' (sample_sgld + project_parameters) | 'sghmc' (extension: momentum with
               friction `friction` in (0,1]; friction = 1 is SGLD)
      window_sampling: 'host' (one window start per chain and step drawn on the host, keyed by
               (seed, global chain id, step) so that a chain's windows do not depend on the rank
               partition; descriptors are re-uploaded) | 'device' (a Philox-keyed kernel rewrites the descriptors in HBM: the
               step is three launches with no host work, and `run(..., graph_steps=K)` replays K
               steps per hipGraph launch)
    """

    def __init__(self, model, observations, parameters, num_chains=None, N=1000, pf="poyiadjis_N",
                 lambduh=None, kernel=None, epsilon=0.1, prior=None, subsequence_length=-1,
                 buffer_length=-1, dtype="f64", seed=0, chain_offset=0, device=None,
                 forward_message=None, partition_style=None, resampling="multinomial",
                 sampler="sgld", friction=0.1, window_sampling="host"):
        if not torch.cuda.is_available():
            raise RuntimeError("ChainEnsemble needs an MI355X (no CPU fallback)")
        Parameters, Prior, Helper = _model_info(model)
        self.model, self.N, self.dtype, self.epsilon = model, int(N), dtype, float(epsilon)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.ctx = _capi.default_context(self.device.index)
        self.helper = Helper(n=1, m=1, forward_message=forward_message)
        self.kernel = self.helper._get_kernel(kernel)
        if pf == "poyiadjis_N":
            self.lambduh = 1.0
        elif pf == "nemeth":
            self.lambduh = 0.95 if lambduh is None else float(lambduh)
        else:
            raise ValueError("ChainEnsemble supports pf = 'poyiadjis_N' | 'nemeth', got {0}".format(pf))
        self.P = _capi.THETA_DIM[model]
        self._Parameters = Parameters
        self.resampling = resampling
        if sampler not in ("sgld", "sghmc"):
            raise ValueError("sampler must be 'sgld' or 'sghmc'")
        self.sampler, self.friction = sampler, float(friction)
        if window_sampling not in ("host", "device"):
            raise ValueError("window_sampling must be 'host' or 'device'")
        self.window_sampling = window_sampling
        self._graphs = {}

        self.segments = None
        if isinstance(observations, (list, tuple)):
            segs = [np.ascontiguousarray(o, dtype=np.float64).reshape(-1) for o in observations]
            if len(segs) == 0 or min(len(o) for o in segs) == 0:
                raise ValueError("every sequence needs at least one observation")
            bounds = np.concatenate([[0], np.cumsum([len(o) for o in segs])]).astype(np.int64)
            self.segments = bounds                    # [K+1] offsets into the concatenated series
            y = np.concatenate(segs)
        else:
            y = np.ascontiguousarray(observations, dtype=np.float64).reshape(-1)
        self.T = y.shape[0]
        if isinstance(parameters, np.ndarray):
            theta0 = np.ascontiguousarray(parameters, dtype=np.float64).reshape(-1, self.P)
            proto = None
        else:
            if num_chains is None:
                raise ValueError("num_chains is required when `parameters` is a Parameters object")
            theta0 = np.tile(parameters.theta(), (int(num_chains), 1))
            proto = parameters
        self.C = theta0.shape[0]
        if prior is None:
            prior = Prior.generate_default_prior(var=1.0 if model == "garch" else 100.0, n=1, m=1)
        self.prior = prior
        self.hyper = prior_hyper(model, prior)
        self.seed, self.chain_offset = int(seed), int(chain_offset)

        S, B = int(subsequence_length), int(buffer_length)
        if self.segments is not None:
            if window_sampling != "host":
                raise NotImplementedError("sequence lists use host-side window sampling")
            if S == -1:
                S = int(np.max(np.diff(self.segments)))      # whole sequences
        elif S == -1 or self.T - S <= 0:
            S = -1
        self.S, self.B = S, (self.T if B == -1 else B)
        self.partition_style = partition_style

        dev = self.device
        th = np.zeros((self.C, _capi.MAX_THETA))
        th[:, :self.P] = theta0
        self.y_dev = torch.from_numpy(y).to(dev)
        self.theta_dev = torch.from_numpy(th).to(dev)
        self.out_dev = torch.zeros((self.C, _capi.OUT_DOUBLES), dtype=torch.float64, device=dev)
        self.step_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.momentum_dev = torch.zeros((self.C, _capi.MAX_THETA), dtype=torch.float64, device=dev)
        self.weights_dev = None
        self._weights_table = None
        if self.segments is not None:
            # one weights block per sequence: rows = window starts (or the single whole-sequence
            # row when the sequence is no longer than S), every row pre-multiplied by
            # T_total / T_sequence -- the Seq sampler's rescaling of a one-sequence gradient
            blocks, offs, off = [], [], 0
            for k in range(len(self.segments) - 1):
                Tk = int(self.segments[k + 1] - self.segments[k])
                scale = self.T / float(Tk)
                if Tk - S <= 0:
                    blk = np.ones((1, Tk)) * scale
                else:
                    blk = np.stack([self._weights_for(st, T=Tk) for st in range(Tk - S + 1)]) * scale
                offs.append(off)
                off += blk.size
                blocks.append(blk.reshape(-1))
            self._seg_weight_offsets = np.array(offs, dtype=np.int64)
            self.weights_dev = torch.from_numpy(np.concatenate(blocks)).to(dev)
        elif S > 0:
            # importance weights depend on the window start only: one resident row per start
            table = np.zeros((self.T - S + 1, S))
            for start in range(self.T - S + 1):
                table[start] = self._weights_for(start)
            self._weights_table = table
            self.weights_dev = torch.from_numpy(table).to(dev)
        pm, pv, _ = self._prior_x(proto, theta0[0])
        self._desc = np.zeros(self.C, dtype=_capi.DEV_PROBLEM_DTYPE)
        d = self._desc
        d["theta"] = self.theta_dev.data_ptr() + np.arange(self.C, dtype=np.uint64) * (8 * _capi.MAX_THETA)
        d["out"] = self.out_dev.data_ptr() + np.arange(self.C, dtype=np.uint64) * (8 * _capi.OUT_DOUBLES)
        d["step_ctr"] = self.step_ctr.data_ptr()
        d["prior_mean"], d["prior_var"], d["lambduh"] = pm, pv, self.lambduh
        d["seed"] = np.uint64(self.seed & 0xFFFFFFFFFFFFFFFF)
        d["stream"] = np.arange(self.C, dtype=np.uint64) + np.uint64(self.chain_offset)
        d["N"] = self.N
        d["smoother"], d["stat"] = _capi.SMOOTHER["nemeth"], _capi.STAT["score"]
        if model == "garch" and self.helper.default_forward_message is None:
            d["flags"] = _capi.FLAG_GARCH_STATIONARY_PRIOR
        if resampling == "systematic":       # extension, see include/pfgrad.h
            if self.N > 1024:
                raise NotImplementedError("systematic resampling is built for N <= 1024")
            d["smoother"] = _capi.SMOOTHER["nemeth_systematic"]
        elif resampling != "multinomial":
            raise ValueError("Unrecognized resampling = {0}".format(resampling))
        sb = self.ctx.scratch_bytes(model, dtype, "device", self.N)
        if sb < 0:
            raise ValueError("N = {0} is above the supported maximum".format(self.N))
        self.scratch_dev = None
        if sb > 0:       # large-N kernel: particle state lives in HBM (L2-resident), one slab per chain
            self.scratch_dev = torch.empty(self.C * sb, dtype=torch.uint8, device=dev)
            d["scratch"] = self.scratch_dev.data_ptr() + np.arange(self.C, dtype=np.uint64) * np.uint64(sb)
        self.steps_done = 0
        if (partition_style or 'uniform') == 'strict' and S > 0 and self.segments is None and self.T % S != 0:
            raise ValueError("S {0} does not evenly divide T {1}".format(S, self.T))     # sgmcmc_sampler.py:1991-1993
        self._set_windows(first=True)
        self.desc_dev = torch.from_numpy(self._desc.view(np.uint8).reshape(self.C, -1)).to(dev)

    # ------------------------------------------------------------------------------------
    def _weights_for(self, start, T=None):
        """weights of random_subsequence_and_weights for a given start ('uniform' style)."""
        S, T = self.S, (self.T if T is None else T)
        style = self.partition_style or 'uniform'
        if style in ('strict', 'naive'):
            return np.ones(S) * T / S
        t = np.arange(start, start + S)
        cap = np.ones_like(t) * min(S, T - S + 1)
        if start + S <= 2 * S:
            covering = np.min(np.array([t + 1, cap]), axis=0)
        elif start >= T - 2 * S - 1:
            covering = np.min(np.array([T - t, cap]), axis=0)
        else:
            covering = np.ones(S) * S
        return np.ones(S, dtype=float) * (T - S + 1) / covering

    def _prior_x(self, proto, theta_row):
        if proto is None:
            proto = self._params_from_theta(theta_row)
        return self.helper._prior_x(None, proto)

    def _params_from_theta(self, th):
        if self.model == "svm":
            return self._Parameters(A=np.eye(1) * th[0], LQinv=np.eye(1) * th[1], LRinv=np.eye(1) * th[2])
        if self.model == "lgssm":
            return self._Parameters(A=np.eye(1) * th[0], C=np.eye(1) * th[1], LQinv=np.eye(1) * th[2],
                                    LRinv=np.eye(1) * th[3])
        return self._Parameters(log_mu=th[0], logit_phi=th[1], logit_lambduh=th[2], LRinv=np.eye(1) * th[3])

    def _host_uniforms(self, salt):
        """[C] uniforms in [0,1) for the step `steps_done`, one per chain, keyed by (seed, GLOBAL chain id,
        step, salt) with a splitmix64 finaliser: a chain's window sequence does not depend on how chains
        are partitioned over ranks (chain_offset / C), nor on what other chains do."""
        M = np.uint64(0xFFFFFFFFFFFFFFFF)
        with np.errstate(over="ignore"):
            x = (np.arange(self.C, dtype=np.uint64) + np.uint64(self.chain_offset)) * np.uint64(0x9E3779B97F4A7C15)
            x ^= np.uint64(self.seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(0xD1B54A32D192ED03)
            x ^= (np.uint64(self.steps_done) * np.uint64(0xBF58476D1CE4E5B9)) ^ (np.uint64(salt) * np.uint64(0x94D049BB133111EB))
            for _ in range(2):
                x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9) & M
                x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB) & M
                x = x ^ (x >> np.uint64(31))
        return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def _set_windows(self, first=False):
        """Full sequence: static descriptors.  Buffered windows: draw one start per chain on the
        host (sgmcmc_sampler.py:259-288) and point y / weights / t1 / tL at it."""
        d = self._desc
        if self.segments is not None:
            # one sequence per chain (np.random.choice(K, 1)), then a window inside it
            K = len(self.segments) - 1
            seg = np.minimum((self._host_uniforms(1) * K).astype(np.int64), K - 1)
            base, Tk = self.segments[seg], self.segments[seg + 1] - self.segments[seg]
            S, B = self.S, self.B
            whole = Tk - S <= 0
            span = np.where(whole, 1, Tk - S + 1)
            if (self.partition_style or 'uniform') == 'strict':
                start = np.where(whole, 0, (self._host_uniforms(2) * np.maximum(Tk // S, 1)).astype(np.int64) * S)
            else:
                start = np.where(whole, 0, (self._host_uniforms(2) * span).astype(np.int64))
            length = np.where(whole, Tk, S)
            left = np.maximum(0, start - B)
            right = np.minimum(Tk, start + length + B)
            d["y"] = self.y_dev.data_ptr() + (base + left).astype(np.uint64) * 8
            d["T"] = right - left
            d["t1"] = start - left
            d["tL"] = start + length - left
            d["weights"] = self.weights_dev.data_ptr() + (self._seg_weight_offsets[seg] + start * length).astype(np.uint64) * 8
            return True
        if self.S == -1:
            if first:
                d["y"] = self.y_dev.data_ptr()
                d["T"], d["t1"], d["tL"] = self.T, 0, self.T
                d["weights"] = 0
            return False
        S, B, T = self.S, self.B, self.T
        if (self.partition_style or 'uniform') == 'strict':
            start = np.minimum((self._host_uniforms(2) * (T // S)).astype(np.int64), T // S - 1) * S
        else:
            start = np.minimum((self._host_uniforms(2) * (T - S + 1)).astype(np.int64), T - S)
        left = np.maximum(0, start - B)
        right = np.minimum(T, start + S + B)
        d["y"] = self.y_dev.data_ptr() + left.astype(np.uint64) * 8
        d["T"] = right - left
        d["t1"] = start - left
        d["tL"] = start + S - left
        d["weights"] = self.weights_dev.data_ptr() + start.astype(np.uint64) * (8 * S)
        return True

    # ------------------------------------------------------------------------------------
    def launch_pf(self, stream=None, traced=False):
        """Enqueue one particle-filter launch for all chains on `stream` (default: torch's
        current stream).  Results land in self.out_dev[C, 8] (score columns, loglik).
        traced=True runs the twin instantiation that honours trace_* / rec_* buffers a caller put into
        the descriptors (tests, diagnostics); the production launch ignores them."""
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        if traced:
            self.ctx.launch_device_traced(self.model, self.kernel, self.dtype, "device",
                                          "nemeth_systematic" if self.resampling == "systematic" else "nemeth",
                                          self.N, self.C, self.desc_dev.data_ptr(), st)
        elif self.resampling == "systematic":
            self.ctx.launch_device_smoother(self.model, self.kernel, self.dtype, "device", "nemeth_systematic",
                                            self.N, self.C, self.desc_dev.data_ptr(), st)
        elif self.lambduh == 1.0:
            # every chain the Poyiadjis O(N) score (NEMETH, lambduh = 1, score): units with a twin specialised to it run that
            self.ctx.launch_device_smoother(self.model, self.kernel, self.dtype, "device", "poyiadjis_n",
                                            self.N, self.C, self.desc_dev.data_ptr(), st)
        else:
            self.ctx.launch_device(self.model, self.kernel, self.dtype, "device", self.N, self.C,
                                   self.desc_dev.data_ptr(), st)

    def launch_update(self, stream=None):
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        if self.sampler == "sghmc":
            self.ctx.sghmc_update_device(self.model, self.C, self.theta_dev.data_ptr(), self.momentum_dev.data_ptr(),
                                         self.out_dev.data_ptr(), self.hyper, self.epsilon, self.friction,
                                         float(self.T), self.seed ^ 0x5DEECE66D, self.chain_offset,
                                         self.step_ctr.data_ptr(), st)
        else:
            self.ctx.sgld_update_device(self.model, self.C, self.theta_dev.data_ptr(), self.out_dev.data_ptr(),
                                        self.hyper, self.epsilon, float(self.T), self.seed ^ 0x5DEECE66D,
                                        self.chain_offset, self.step_ctr.data_ptr(), st)

    def launch_windows(self, stream=None):
        """Device-side window sampling (window_sampling='device'): rewrite y / T / t1 / tL / weights of
        every descriptor for the step *step_ctr is at.  No-op for full-sequence chains."""
        if self.S == -1:
            return
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        self.ctx.sample_windows_device(
            self.C, self.desc_dev.data_ptr(), self.y_dev.data_ptr(),
            self.weights_dev.data_ptr() if self.weights_dev is not None else 0, self.T, self.S, self.B,
            (self.partition_style or 'uniform') == 'strict', self.seed ^ 0x2545F4914F6CDD1D, self.chain_offset,
            self.step_ctr.data_ptr(), st)

    def _enqueue_step(self):
        if self.window_sampling == "device":
            self.launch_windows()
        elif self.steps_done > 0 and self._set_windows():
            self.desc_dev.copy_(torch.from_numpy(self._desc.view(np.uint8).reshape(self.C, -1)),
                                non_blocking=True)
        self.launch_pf()
        self.launch_update()

    def step(self, num_steps=1):
        """num_steps x (sample_sgld + project_parameters) for every chain.  Asynchronous."""
        for _ in range(num_steps):
            self._enqueue_step()
            self.steps_done += 1

    def _graph(self, K):
        """A hipGraph of K whole steps (window sampling, particle filter, update -- 3K kernel nodes).
        Every input that changes between steps (parameters, descriptors, RNG step counter) lives
        in HBM and is advanced by the kernels themselves, so replaying the graph IS running K
        more steps; it is bitwise the same computation as K eager steps."""
        if self.S != -1 and self.window_sampling != "device":
            raise ValueError("graph capture needs window_sampling='device' (or full-sequence chains): "
                             "host-side window sampling cannot be replayed")
        g = self._graphs.get(K)
        if g is None:
            # one eager step first (code-object load, LDS-size attributes), undone afterwards so
            # that building the graph does not advance the chains
            snap = [t.clone() for t in (self.theta_dev, self.momentum_dev, self.step_ctr, self.desc_dev)]
            self._enqueue_step()
            self.synchronize()
            for t, c in zip((self.theta_dev, self.momentum_dev, self.step_ctr, self.desc_dev), snap):
                t.copy_(c)
            self.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(K):
                    self._enqueue_step()
            self._graphs[K] = g
        return g

    def run(self, num_steps, thin=1, graph_steps=0):
        """num_steps steps, keeping every `thin`-th state: returns ndarray [num_steps // thin, C, P].
        Samples are staged in HBM and copied to the host once at the end.  graph_steps = K > 0
        replays a captured hipGraph of K steps per launch (K must divide `thin`); worth it when a
        step is launch-bound (short buffered windows, few chains)."""
        keep = num_steps // thin
        buf = torch.empty((max(keep, 1), self.C, self.P), dtype=torch.float64, device=self.device)
        k = 0
        K = int(graph_steps)
        if K > 0:
            if thin % K != 0:
                raise ValueError("graph_steps must divide thin")
            g = self._graph(K)
            it = 0
            while it < num_steps:
                if num_steps - it >= K:
                    g.replay()
                    self.steps_done += K
                    it += K
                else:
                    self.step(1)
                    it += 1
                if it % thin == 0 and k < keep:
                    buf[k].copy_(self.theta_dev[:, :self.P])
                    k += 1
            return buf[:keep].cpu().numpy()
        for it in range(1, num_steps + 1):
            self.step(1)
            if it % thin == 0 and k < keep:
                buf[k].copy_(self.theta_dev[:, :self.P])
                k += 1
        return buf[:keep].cpu().numpy()

    # -- checkpoint / resume (the reference checkpoints parameters with joblib around its fit loop,
    #    svm/driver.py:362-408; here the whole ensemble state is a few small arrays) -------------
    def state_dict(self):
        self.synchronize()
        return dict(theta=self.theta_dev.cpu().numpy(), momentum=self.momentum_dev.cpu().numpy(),
                    step_ctr=int(self.step_ctr.item()), steps_done=int(self.steps_done),
                    seed=self.seed, chain_offset=self.chain_offset,
                    model=self.model, N=self.N, C=self.C)

    def load_state_dict(self, state):
        for key in ("model", "N", "C", "seed", "chain_offset"):
            if state[key] != getattr(self, key):
                raise ValueError("checkpoint {0} = {1} does not match the ensemble ({2})".format(
                    key, state[key], getattr(self, key)))
        self.theta_dev.copy_(torch.from_numpy(np.ascontiguousarray(state["theta"])))
        self.momentum_dev.copy_(torch.from_numpy(np.ascontiguousarray(state["momentum"])))
        self.step_ctr.fill_(int(state["step_ctr"]))
        self.steps_done = int(state["steps_done"])       # host window draws are keyed by (seed, chain, steps_done)
        self.synchronize()

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # -- measurement hooks ---------------------------------------------------------------------
    def enable_stamps(self):
        """Point every descriptor at a [C, 16] uint64 stamp record (pfg_dev_problem.stamps): the PF
        kernel's wave 0 then writes s_memtime / s_memrealtime at its start and end (two scalar
        instructions outside the T-loop)."""
        self.stamps_dev = torch.zeros((self.C, _capi.STAMP_WORDS), dtype=torch.int64, device=self.device)
        self._desc["stamps"] = self.stamps_dev.data_ptr() + np.arange(self.C, dtype=np.uint64) * np.uint64(8 * _capi.STAMP_WORDS)
        self.desc_dev.copy_(torch.from_numpy(self._desc.view(np.uint8).reshape(self.C, -1)))
        self.synchronize()

    def kernel_clock(self):
        """From the stamps of the latest PF launch: (median in-kernel shader clock in GHz, median
        workgroup lifetime in shader cycles, per-phase cycle sums [10] or None).  The phase sums are
        filled by diagnostic builds only (-DPFG_PHASE_STAMPS)."""
        st = self.stamps_dev.cpu().numpy().astype(np.uint64)
        cyc = (st[:, 2] - st[:, 0]).astype(np.float64)
        real = (st[:, 3] - st[:, 1]).astype(np.float64)          # 100 MHz ticks
        ok = real > 0
        ghz = float(np.median(cyc[ok] / real[ok] * 0.1)) if ok.any() else float("nan")
        phases = st[:, 4:14].sum(axis=0).astype(np.float64)
        return ghz, float(np.median(cyc[ok])) if ok.any() else float("nan"), (phases if phases.sum() > 0 else None)

    # ------------------------------------------------------------------------------------
    def theta(self):
        """Current raw parameters of all chains, ndarray [C, P] (synchronises)."""
        return self.theta_dev[:, :self.P].cpu().numpy()

    def last_gradient_statistics(self):
        """[C, h] score estimates and [C] log-likelihood estimates of the latest PF launch."""
        out = self.out_dev.cpu().numpy()
        return out[:, :_capi.STAT_DIM[self.model]], out[:, 4]

    def parameters_list(self):
        return [self._params_from_theta(th) for th in self.theta()]

    def gather_samples(self):
        """All ranks' current samples [world*C, P] on every rank, in global chain order: the one
        collective of the multi-GPU path (RCCL all_gather over xGMI; a few KB, latency-bound)."""
        from . import distributed
        return distributed.gather_samples(self.theta_dev[:, :self.P].contiguous())
