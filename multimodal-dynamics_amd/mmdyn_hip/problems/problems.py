"""Problem layer: the caller of the hot path, restated for synthetic / in-memory batches.

Mirrors the parts of /root/reference/mmdyn/pytorch/problems/problems.py that drive the model:
``Problem`` (ctor flags, ``set_optimizer`` :130-138, ``_train_epoch`` step body :148-156,
``_test_epoch`` :173-191, ``train`` :193-210, ``_anneal_KL`` :212-216), ``Reconstruction``
(``set_model`` :367-389, ``_elbo_loss`` :401-419, ``_mvae_elbo_loss`` :421-458, ``_evaluate_mvae``
:473-546, best-validation checkpoint :580-586), ``SeqModeling`` (``parse_input`` :634-673,
``_evaluate_model`` :683-716) and ``DynModeling.parse_input`` (:765-803).

Out of scope here: the PNG/json dataset reader, TensorBoard image/figure logging,
``Regression``.  Batches come from ``SyntheticVisuoTactile`` (random 64x64 visual + tactile + 7-DoF pose,
the BASELINE.json workload) or from any iterable yielding the reference's ``(data_input, data_target)``
lists.

Two execution modes for cnn-mvae training:
  * ``fused=True`` (default): :class:`mmdyn_hip.engine.MVAEStep`, the restructured kernel schedule;
  * ``fused=False``: the reference's own schedule -- seven ``model(...)`` calls and ``loss.backward()``
    through the autograd Functions of :mod:`mmdyn_hip.models.functional` (drop-in semantics, including
    the decoder passes whose output is discarded).
"""
import json
import os
import pickle
import time
from collections import defaultdict
from datetime import datetime
from pathlib import Path

import torch

from .. import config, ops
from ..engine import MVAEStep
from ..models import functional as Fn
from ..models.models import setup_model


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr) semantics (betas (0.9, 0.999), eps 1e-8, no weight decay) on the mmdyn_adam_step
    kernel, one launch per parameter tensor.  Used by the un-fused (reference-schedule) mode."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                    st["state"] = torch.zeros(3, dtype=torch.float64, device=p.device)
                b1, b2 = group["betas"]
                ops.B.adam_step(p.data.view(-1), p.grad.contiguous().view(-1), st["m"].view(-1), st["v"].view(-1),
                                st["state"], group["lr"], b1, b2, group["eps"], 1.0)


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum=0.9, weight_decay=5e-4) -- the reference's SGD option (problems.py:132-136) -- on
    the mmdyn_sgd_step kernel."""

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=5e-4):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = not st
                if first:
                    st["buf"] = torch.zeros_like(p)
                ops.B.sgd_step(p.data.view(-1), p.grad.contiguous().view(-1), st["buf"].view(-1), group["lr"],
                               group["momentum"], group["weight_decay"], 1.0, first)


class SyntheticVisuoTactile:
    """Iterable of ``(data_input, data_target)`` in the reference's collated list format
    (datasets.py:395-404): [visual, tactile, pose, available_modals(, shock)] / [visual, tactile, pose, mask],
    frames of ``seq_length`` per sequence folded into the batch dimension.  Values are U[0,1) like
    ``ToTensor`` images and min-max normalised poses (datasets.py:23-31, 407-408)."""

    def __init__(self, n_batches, batchsize, seq_length=1, device="cpu", seed=1234, size=64, shock_dim=0):
        self.n_batches, self.batchsize, self.seq_length = n_batches, batchsize, seq_length
        self.device, self.seed, self.size, self.shock_dim = device, seed, size, shock_dim

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        n = self.batchsize * self.seq_length
        for _ in range(self.n_batches):
            def draw():
                return [torch.rand(n, 3, self.size, self.size, generator=g), torch.rand(n, 3, self.size, self.size, generator=g),
                        torch.rand(n, 7, generator=g)]
            d, t = draw(), draw()
            d.append(torch.ones(n, 2))
            if self.shock_dim:
                d.append(torch.rand(n, self.shock_dim, generator=g))      # the shock force (datasets.py:395-404)
            t.append(torch.ones(n, 1, self.size, self.size))
            yield [x.to(self.device) for x in d], [x.to(self.device) for x in t]


class Problem:
    def __init__(self, problem_args, log_dir=None, load_dataset=None, train_loader=None, test_loader=None,
                 seq_length=1, fused=True):
        self._model = None
        self._optimizer = None
        self._step = None
        self._best_loss = float("inf")
        self._logger_dict = defaultdict(list)
        self.parameters = vars(problem_args) if not isinstance(problem_args, dict) else dict(problem_args)
        self._cross_modal = self.parameters['input_type'] == 'visuotactile'
        self._kl_weight = self.parameters['kl_weight']
        self._pose_multiplier = self.parameters['pose_multiplier']
        self._conditional = self.parameters['conditional']
        self._categorical_conditions = None
        self._condition_dim = 0
        self._seq_length = seq_length
        self._fused = fused
        self.train_loader, self.test_loader = train_loader, test_loader
        use_gpu = torch.cuda.is_available() and not self.parameters['no_cuda']
        if not use_gpu and ops.B.name == "hip":
            raise RuntimeError("mmdyn_hip needs a ROCm GPU: there is no CPU path (run the reference for --no-cuda)")
        self._device = torch.device('cuda' if use_gpu else 'cpu')
        assert (self.parameters['input_type'] in config.INPUT_TYPES), "Input type is not implemented"
        if self.train_loader is None and self.parameters.get('dataset_path'):
            self.set_dataset()
        if log_dir:
            self.load_dir(log_dir)
        else:
            self.set_dir()
        self._set_problem()

    def set_dataset(self):
        """problems.py:110-125: the on-disk dataset at --dataset-path, decoded on the GPU (utils/datasets.py)."""
        from ..utils.datasets import dataset_setup
        self._input_size = (int(self.parameters.get('image_size', 64)),) * 2      # reference: (64, 64), problems.py:111
        self._n_channels = 3
        self.dataset_dict = dataset_setup(self.parameters['dataset_path'], self.parameters['problem_type'],
                                          input_size=self._input_size, batchsize=self.parameters['batchsize'],
                                          shuffle=True, device=self._device)
        self.train_dataset, self.test_dataset = self.dataset_dict['train_dataset'], self.dataset_dict['test_dataset']
        self.train_loader, self.test_loader = self.dataset_dict['train_loader'], self.dataset_dict['test_loader']
        self._seq_length = self.dataset_dict['seq_length']
        print(len(self.train_dataset), len(self.test_dataset))

    def _set_problem(self):
        self.set_model()
        self.set_criterion()
        self.set_optimizer()

    def load_dir(self, log_dir):
        self._log_dir = log_dir
        self._checkpoint_dir = self._log_dir + '/checkpoint/'
        for d in (self._log_dir, self._checkpoint_dir):
            Path(d).mkdir(parents=True, exist_ok=True)

    def set_dir(self):
        date = datetime.now().strftime("_%Y_%m_%d_%H_%M_%S")
        self.load_dir('./logs/' + self.parameters['save_name'] + date)

    def set_model(self):
        raise NotImplementedError

    def set_criterion(self):
        raise NotImplementedError

    def set_optimizer(self):
        assert (self.parameters['optimizer'] in config.OPTIMIZERS), "loss name not implemented in Problem"
        if self.parameters['optimizer'] == 'SGD':      # reference: momentum 0.9, weight decay 5e-4 (problems.py:132-136)
            self._optimizer = FusedSGD(self._model.parameters(), lr=self.parameters['lr'], momentum=0.9, weight_decay=5e-4)
            return
        # the fused step takes the dict-shaped inputs of seq / dyn modeling (with the loss mask of --mask-loss when the model has
        # no pose term, and the shock condition of --conditional); everything else (plain reconstruction, cnn-vae, regressor,
        # --mask-loss with --use-pose, which fails in the reference too: problems.py:445-447) runs the module path
        # (decided from the model that was actually built: 'cnn-mvae' with a single-modality --input-type is a plain VAE)
        from ..models.vae import MVAE
        use_engine = (self._fused and isinstance(self._model, MVAE) and self._cross_modal
                      and isinstance(self, SeqModeling)
                      and not (self.parameters.get('mask_loss') and self.parameters.get('use_pose')))
        # fp32x3 (fp32 storage and results, the GEMMs on the bf16 matrix cores through the exact three-term operand split) is the
        # fused step's default arithmetic; the module path below always computes on the native fp32 matrix cores
        precision = self.parameters.get('precision', 'fp32x3')
        if use_engine:
            self._step = MVAEStep(self._model, lr=self.parameters['lr'], pose_multiplier=self._pose_multiplier,
                                  precision=precision, exact_running_stats=bool(self.parameters.get('exact_running_stats')))
        elif precision not in ('fp32', 'fp32x3'):
            raise ValueError("--precision %s is a mode of the fused cnn-mvae step; this configuration runs the module "
                             "path, which computes in fp32" % precision)
        # the module path's optimiser: also what a batch the fused step does not take falls back to
        self._optimizer = FusedAdam(self._model.parameters(), lr=self.parameters['lr'])

    def _anneal_KL(self, epoch):
        if epoch < self.parameters['annealing_epochs']:
            self._kl_weight = (epoch + 1) / self.parameters['annealing_epochs']
        else:
            self._kl_weight = 1

    # ---- epoch loops -----------------------------------------------------------------------------
    def _train_epoch(self, epoch):
        self._model.train()
        train_loss, n = 0.0, 0
        perf = defaultdict(float)
        dev_loss = dev_acc = None        # fused path: loss and per-pass sums accumulate ON THE DEVICE, read once per epoch
        dev_n = dev_rows = 0             # (the reference's loss.item() per step, problems.py:156, would stall the replay)
        try:
            n_batches = len(self.train_loader)
        except TypeError:
            n_batches = -1
        for batch_idx, (data_input, data_target) in enumerate(self.train_loader):
            inputs, targets = self.parse_input(data_input, data_target)
            if self._step is not None and self._fused_applicable(inputs):
                # on the GPU the step is replayed from HIP graphs (captured once per batch shape; the annealed KL weight
                # is read from device memory); the emulation has no graphs
                run = self._step.train_step_graphed if self._device.type == 'cuda' else self._step.train_step
                # Checkpoint fidelity: the LAST step of an epoch -- what the epoch's checkpoint sees -- also runs the image
                # decoders on the subset passes whose reconstruction the reference computes and discards, so their BatchNorm
                # running buffers receive the reference's 7 (3) EMA updates of that step, in pass order (eager launches:
                # the captured graphs hold the 4-pass schedule).  With momentum 0.1 those updates carry 1 - 0.9**7 = 52 %
                # of a buffer's weight; tests/test_model_emu.py measures the residual against the reference's buffers.
                # --exact-running-stats does it on every step (identical buffers, ~1.4x the step time).
                last_exact = (batch_idx == n_batches - 1 and not self._step.exact_running_stats and
                              self.parameters.get('exact_last_step', True))
                if last_exact:
                    self._step.exact_running_stats, run = True, self._step.train_step
                try:
                    loss = run(*self._fused_io(inputs, targets), self._kl_weight, loss_mask=self._fused_mask(targets),
                               condition=inputs.get('shock') if self._conditional else None)
                finally:
                    if last_exact:
                        self._step.exact_running_stats = False
                if dev_loss is None:
                    dev_loss, dev_acc = torch.zeros_like(loss, dtype=torch.float64), torch.zeros_like(self._step.acc)
                dev_loss += loss.detach().to(torch.float64)
                dev_acc += self._step.acc
                dev_n += 1
                dev_rows += self._step.last['means'].shape[0]
                continue
            else:
                self._optimizer.zero_grad()
                outputs, loss = self._evaluate_model(inputs, targets)
                loss.backward()
                self._optimizer.step()
            train_loss += float(loss.detach()) if torch.is_tensor(loss) else float(loss)
            n += 1
            for k, v in outputs.get('perf_measure', {}).items():
                perf[k] += v
        if dev_n:
            train_loss += float(dev_loss.sum())
            for k, v in self._fused_perf(dev_acc.cpu(), dev_rows / dev_n).items():
                perf[k] += v          # (sums over the epoch's steps / rows per step = the sum of the per-step means)
            n += dev_n
        self._logger_dict['Loss/train_epoch'].append(train_loss / max(n, 1))
        self._logger_dict['KL_annealing/train_epoch'].append(self._kl_weight)
        for k, v in perf.items():
            self._logger_dict['Perf_measure_train/' + k].append(v / max(n, 1))
        return dict(perf)

    def _test_epoch(self, epoch):
        self._model.train()          # sic: the reference validates in train mode (problems.py:174)
        val_loss, n = 0.0, 0
        perf = defaultdict(float)
        with torch.no_grad():
            for batch_idx, (data_input, data_target) in enumerate(self.test_loader):
                inputs, targets = self.parse_input(data_input, data_target)
                if self._step is not None and self._fused_applicable(inputs):
                    loss = self._step.eval_step(*self._fused_io(inputs, targets), self._kl_weight,
                                                loss_mask=self._fused_mask(targets),
                                                condition=inputs.get('shock') if self._conditional else None)
                    outputs = {'perf_measure': self._fused_perf()}
                else:
                    outputs, loss = self._evaluate_model(inputs, targets)
                val_loss += float(loss)
                n += 1
                for k, v in outputs.get('perf_measure', {}).items():
                    perf[k] += v
        self._logger_dict['Loss/validation_epoch'].append(val_loss / max(n, 1))
        for k, v in perf.items():
            self._logger_dict['Perf_measure_validation/' + k].append(v / max(n, 1))
        if val_loss < self._best_loss:      # best-validation checkpoint, same dict keys as problems.py:580-586
            state = {'model': self._model.state_dict(), 'loss': val_loss, 'epoch': epoch}
            torch.save(state, self._checkpoint_dir + '/epoch_' + str(epoch) + '.ckpt')
            if self._step is not None:
                # the checkpoint keeps the reference's three keys (problems.py:751-757); how the image decoders' BatchNorm
                # running buffers in 'model' were produced goes next to it
                mode = ('exact' if self._step.exact_running_stats else
                        'exact-on-last-step-of-epoch' if self.parameters.get('exact_last_step', True) else 'live-passes-only')
                with open(self._checkpoint_dir + '/bn_running_stats_mode.txt', 'w') as f:
                    f.write(mode + '\n')
            self._best_loss = val_loss
        return dict(perf)

    def train(self, save=True):
        perf = {}
        for epoch in range(self.parameters['num_epochs']):
            t0 = time.time()
            self._anneal_KL(epoch)
            self._train_epoch(epoch)
            if self.test_loader is not None:
                perf = self._test_epoch(epoch)
            row = {k: v[-1] for k, v in self._logger_dict.items() if v}
            row.update(epoch=epoch, seconds=time.time() - t0)
            with open(os.path.join(self._log_dir, 'scalars.jsonl'), 'a') as f:
                f.write(json.dumps(row) + "\n")
            print('Epoch: %d  %s' % (epoch, json.dumps(row)))
        if save:
            with open(os.path.join(self._log_dir, 'results.pkl'), 'wb') as f:
                pickle.dump(dict(self._logger_dict), f)
        return perf

    # ---- fused-engine plumbing -------------------------------------------------------------------
    def _fused_applicable(self, inputs):
        return isinstance(inputs, dict) and isinstance(inputs.get('model_input'), list)

    def _fused_mask(self, targets):
        return targets['loss_mask'] if self.parameters.get('mask_loss') else None

    def _fused_io(self, x, targets):
        if self.parameters['use_pose']:
            return x['model_input'] + x['input_object_pose'], targets['target_output'] + targets['target_object_pose']
        return x['model_input'], targets['target_output']

    def _fused_perf(self, acc=None, B=None):
        """Mean BCE / MSE of the single-modality passes (problems.py:499-503, 534-535) from the engine's per-pass sums
        (``acc``: those sums, or their total over several steps of ``B`` rows each)."""
        st = self._step
        B = st.last['means'].shape[0] if B is None else B
        acc = st.acc.cpu() if acc is None else acc
        npx = st.last['recon_x'][0][0].numel()              # 3 * S * S (12288 for the reference's 64 x 64)
        row = 3 if st.last.get('masked') else 0            # the perf measures are unmasked sums (problems.py:495-505)
        out = {'visual': float(acc[row, 1]) / (B * npx), 'tactile': float(acc[row, 2]) / (B * npx)}
        if st.use_pose:
            out['pose'] = float(acc[1, 6]) / (B * 7)
        return out

    @property
    def log_dir(self):
        return self._log_dir

    @property
    def model(self):
        return self._model

    @property
    def checkpoint_dir(self):
        return self._checkpoint_dir

    @property
    def num_epochs(self):
        return self.parameters['num_epochs']

    @property
    def input_type(self):
        return self.parameters['input_type']

    @property
    def condition_dim(self):
        return self._condition_dim


class Regression(Problem):
    """Baseline regressing the object pose from one image modality (problems.py:263-359).  The reference's
    ``set_model`` passes a ``condition_dim`` keyword its Regressor does not take (problems.py:275 vs
    models.py:30) and so cannot run as shipped; here the shock dimension goes in as ``num_classes``, the
    argument the Regressor actually sizes its conditional input with (models.py:37)."""

    def set_model(self):
        self._categorical_conditions = False
        self._condition_dim = int(getattr(self.train_loader, 'shock_dim', 0) or 0)
        if self._conditional and not self._condition_dim:
            raise ValueError("--conditional needs a dataset that carries the shock force (data[4])")
        self._model = setup_model(self.parameters['model_name'], out_dim=7, conditional=self._conditional,
                                  num_classes=self._condition_dim)
        self._model.to(self._device)

    def set_criterion(self):
        self._criterion = None                # MSELoss(reduction='sum') == Fn.MSESumFn

    def parse_input(self, data, target):
        """problems.py:290-316: one image modality in, the target pose (target[2]) out, every l-th frame."""
        l, dev = self._seq_length, self._device
        if not isinstance(data, list):
            mi, to = data.to(dev), target.to(dev)
        elif len(data) == 1:
            mi, to = data[0].to(dev), target[0].to(dev)
        else:
            i = {'visual': 0, 'tactile': 1}[self.parameters['input_type']]
            mi, to = data[i][::l].to(dev), target[2][::l].to(dev)
        shock = data[4][::l].to(dev) if isinstance(data, list) and len(data) > 4 else None
        return {'model_input': mi, 'shock': shock}, to

    def _evaluate_model(self, inputs, targets, **kwargs):
        out = self._model(inputs['model_input'], inputs['shock']) if self._conditional \
            else self._model(inputs['model_input'])
        out = out.view(targets.size())
        loss = Fn.MSESumFn.apply(out, targets.contiguous())
        return {'outputs': out, 'perf_measure': {'pose': float(loss.detach()) / targets.numel()}}, loss

    def _sample(self, n=50):
        pass


class Reconstruction(Problem):

    def set_model(self):
        self._set_condition_dim()
        model_kwargs = {'condition_dim': self._condition_dim, 'input_dim': int(self.parameters.get('image_size', 64)) ** 2,
                        'architecture': self.parameters['model_name'].split('-')[0],
                        'conditional': self._conditional, 'categorical_conditions': self._categorical_conditions,
                        'latent_size': self.parameters.get('latent_size', 256)}
        if 'mvae' in self.parameters['model_name']:
            model_kwargs['use_pose'] = self.parameters['use_pose']
        self._model = setup_model(self.parameters['model_name'], cross_modal=self._cross_modal, **model_kwargs)
        self._model.to(self._device)

    def _set_condition_dim(self):
        self._categorical_conditions = False
        self._condition_dim = 0

    def set_criterion(self):
        self._criterion = self._mvae_elbo_loss if 'mvae' in self.parameters['model_name'] else self._elbo_loss

    def _elbo_loss(self, recon_x, x, means, log_var, loss_mask=None, reduce=None, reduction='sum'):
        """(BCE_sum + kl_weight * KL) / B for VAE / CVAE (problems.py:401-419, reduce=None branch)."""
        if reduce is not None or reduction != 'sum':
            raise NotImplementedError("mmdyn_hip: per-sample (reduce) losses are never used by main.py")
        batch_size = x.size(0)
        KLD = Fn.KLFn.apply(means, log_var)
        BCE = Fn.BCEWithLogitsSumFn.apply(recon_x.view(x.size()), x, loss_mask)
        return (BCE + self._kl_weight * KLD) / batch_size

    def _mvae_elbo_loss(self, recon_x, x, means, log_var, loss_mask=None, reduce=None, reduction='sum'):
        """Sum over modalities of BCE (images) / pose_multiplier * MSE (pose) + kl_weight * KL, over B
        (problems.py:421-458, reduce=None branch)."""
        if reduce is not None or reduction != 'sum':
            raise NotImplementedError("mmdyn_hip: per-sample (reduce) losses are never used by main.py")
        assert len(recon_x) == len(x)
        batch_size = x[0].size(0)
        recon_error = 0
        kl_divergence = Fn.KLFn.apply(means, log_var)
        for i in range(len(recon_x)):
            if len(recon_x[i].size()) > 2:
                e = Fn.BCEWithLogitsSumFn.apply(recon_x[i].view(x[i].size()), x[i], loss_mask)
            else:
                if loss_mask is not None:
                    raise ValueError("loss_mask is image-shaped and cannot multiply the (B, 7) pose term "
                                     "(the reference raises here too: problems.py:445-447)")
                e = self._pose_multiplier * Fn.MSESumFn.apply(recon_x[i], x[i])
            recon_error = recon_error + e
        return (recon_error + self._kl_weight * kl_divergence) / batch_size

    def _evaluate_model(self, x, targets, **kwargs):
        if 'mvae' in self.parameters['model_name']:
            return self._evaluate_mvae(x=x, targets=x)
        recon_x, means, log_var = self._model(x)
        loss = self._criterion(recon_x, x, means, log_var)
        return {'recon_x': recon_x, 'means': means, 'log_var': log_var}, loss

    def _evaluate_mvae(self, x, targets, loss_mask=None, reduce=None, reduction='sum', condition=None):
        """The reference's 3- or 7-subset schedule, one full model call per subset (problems.py:473-546)."""
        assert isinstance(x, list) and isinstance(targets, list)
        kw = dict(loss_mask=loss_mask, reduce=reduce, reduction=reduction)
        m = self._model
        v_joint, t_joint, _, means, log_var = m([x[0], x[1]], condition=condition)
        loss = self._mvae_elbo_loss([v_joint, t_joint], [targets[0], targets[1]], means, log_var, **kw)
        v_only, _, _, means, log_var = m([x[0], None], condition=condition)
        loss = loss + self._mvae_elbo_loss([v_only], [targets[0]], means, log_var, **kw)
        _, t_only, _, means, log_var = m([None, x[1]], condition=condition)
        loss = loss + self._mvae_elbo_loss([t_only], [targets[1]], means, log_var, **kw)
        with torch.no_grad():
            n_img = targets[0].numel()
            perf = {'visual': float(Fn.BCEWithLogitsSumFn.apply(v_only, targets[0], None)) / n_img,
                    'tactile': float(Fn.BCEWithLogitsSumFn.apply(t_only, targets[1], None)) / n_img}
        if self.parameters['use_pose']:
            v_joint, t_joint, p_joint, means, log_var = m([x[0], x[1]], pose=x[2], condition=condition)
            loss = loss + self._mvae_elbo_loss([v_joint, t_joint, p_joint], [targets[0], targets[1], targets[2]],
                                               means, log_var, **kw)
            v_r, _, p_r, means, log_var = m([x[0], None], pose=x[2], condition=condition)
            loss = loss + self._mvae_elbo_loss([v_r, p_r], [targets[0], targets[2]], means, log_var, **kw)
            _, t_r, p_r, means, log_var = m([None, x[1]], pose=x[2], condition=condition)
            loss = loss + self._mvae_elbo_loss([t_r, p_r], [targets[1], targets[2]], means, log_var, **kw)
            _, _, p_only, means, log_var = m([None, None], pose=x[2], condition=condition)
            loss = loss + self._mvae_elbo_loss([p_only], [targets[2]], means, log_var, **kw)
            with torch.no_grad():
                perf['pose'] = float(Fn.MSESumFn.apply(p_only, targets[2])) / targets[2].numel()
            recon = [v_joint, t_joint, p_joint]
        else:
            recon = [v_joint, t_joint]
        return {'recon_x': recon, 'means': means, 'log_var': log_var, 'perf_measure': perf}, loss

    def _sample(self, n=50):
        with torch.no_grad():
            return self._model.inference(n=n)

    def parse_input(self, data, target):
        if not isinstance(data, list):
            return data.to(self._device), target.to(self._device)
        it = self.parameters['input_type']
        if it == 'visual':
            mi = data[0].to(self._device)
        elif it == 'tactile':
            mi = data[1].to(self._device)
        else:
            mi = [data[0].to(self._device), data[1].to(self._device)]
        return mi, target.to(self._device) if torch.is_tensor(target) else target


class SeqModeling(Reconstruction):

    def _set_condition_dim(self):
        """condition_dim = the shock-force dimension of the dataset (problems.py:675-681)."""
        self._categorical_conditions = False
        self._condition_dim = int(getattr(self.train_loader, 'shock_dim', 0) or 0)
        if self._conditional and not self._condition_dim:
            raise ValueError("--conditional needs a dataset that carries the shock force (data[4])")

    def parse_input(self, data, target):
        """First frame of every sequence -> input, dataset's final frame -> target ([::l], problems.py:634-673)."""
        l = self._seq_length
        dev = self._device
        it = self.parameters['input_type']
        if not isinstance(data, list):
            return {'model_input': data.to(dev), 'shock': None}, {'target_output': target.to(dev), 'loss_mask': None}
        idx = {'visual': [0], 'tactile': [1], 'visuotactile': [0, 1]}[it]
        mi = [data[i][::l].to(dev) for i in idx]
        to = [target[i][::l].to(dev) for i in idx]
        if len(idx) == 1:
            mi, to = mi[0], to[0]
        if len(data) > 2:
            pose_in, avail = [data[2][::l].to(dev)], data[3][::l].to(dev)
            pose_t, mask = [target[2][::l].to(dev)], target[3][::l].to(dev)
            shock = data[4][::l].to(dev) if len(data) > 4 else None
        else:
            pose_in = avail = pose_t = mask = shock = None
        return ({'model_input': mi, 'input_object_pose': pose_in, 'input_available_modals': avail, 'shock': shock},
                {'target_output': to, 'target_object_pose': pose_t, 'loss_mask': mask})

    def _evaluate_model(self, x, targets, reduction='sum', reduce=None, **kwargs):
        loss_mask = targets['loss_mask'] if self.parameters['mask_loss'] else None
        if 'mvae' in self.parameters['model_name']:
            xs, ts = self._fused_io(x, targets)
            return self._evaluate_mvae(x=xs, targets=ts, loss_mask=loss_mask, reduce=reduce, reduction=reduction,
                                       condition=x.get('shock'))
        if self._conditional:
            recon_x, means, log_var = self._model(x['model_input'], x['shock'])
        else:
            recon_x, means, log_var = self._model(x['model_input'])
        loss = self._elbo_loss(recon_x, targets['target_output'], means, log_var, loss_mask=loss_mask, reduce=reduce,
                               reduction=reduction)
        with torch.no_grad():
            tgt = targets['target_output']
            measure = float(Fn.BCEWithLogitsSumFn.apply(recon_x, tgt, None)) / tgt.numel()
        return {'recon_x': recon_x, 'means': means, 'log_var': log_var,
                'perf_measure': {self.parameters['input_type']: measure}}, loss


class DynModeling(SeqModeling):

    def set_dataset(self):
        """The reader reports seq_length only when it has just compiled the tree (datasets.py:88-93 vs 167-171; with
        an existing pickle the reference gets None, and its ``l-1::l`` below raises).  The one-step predictor needs the
        real frame count, so it is read off the data in that case."""
        super().set_dataset()
        if self._seq_length is None:
            self._seq_length = self.train_dataset.frames_per_item

    def parse_input(self, data, target):
        """One-step-ahead targets on flat [B*L, ...] frames (problems.py:765-803): roll by -1, the last frame
        of each sequence takes the dataset's final target (images only; the pose target keeps the plain
        roll, wrap-around included, exactly as the reference does)."""
        l = self._seq_length
        dev = self._device
        idx = {'visual': [0], 'tactile': [1], 'visuotactile': [0, 1]}[self.parameters['input_type']]
        mi, to = [], []
        for i in idx:
            mi.append(data[i].to(dev))
            tgt = torch.roll(data[i], -1, dims=0).to(dev)
            tgt[l - 1::l] = target[i][l - 1::l].to(dev)
            to.append(tgt)
        if len(idx) == 1:
            mi, to = mi[0], to[0]
        shock = data[4].to(dev) if len(data) > 4 else None
        return ({'model_input': mi, 'input_object_pose': [data[2].to(dev)], 'input_available_modals': data[3].to(dev),
                 'shock': shock},
                {'target_output': to, 'target_object_pose': [torch.roll(data[2], -1, dims=0).to(dev)],
                 'loss_mask': target[3].to(dev)})
