"""What the stage-4 script does AROUND the hot path: joint statistics -> the frozen scale_in / scale_out
layers and the initial feedback vectors (reference src/bin/train_gru_cyclevae_gauss_batch.py:296-299, :344-347, :357-359;
statistics as src/bin/calc_stats_vc_joint.py:83-127 computes them with sklearn's StandardScaler), and the checkpoint dictionary
(:152-167 save, :348-355 / :378-379 / :653-655 resume) -- same keys, same file name, loadable by the reference and vice versa.
"""
import os

import numpy as np
import torch


def joint_stats(arrays):
    """mean_ / scale_ of sklearn.preprocessing.StandardScaler after partial_fit over `arrays` ([frames, dim] each), as
    calc_stats_vc_joint.py:83-127 stores them under /mean_feat_org_lf0_jnt and /scale_feat_org_lf0_jnt: per-dimension mean and
    POPULATION standard deviation over all frames, a zero deviation replaced by 1.  float64.  ONE pass (any iterable, a generator
    that reads one file at a time included): per-array mean and sum of squared deviations merged by Chan's update, the scheme of
    sklearn's partial_fit -- no cancellation, nothing kept but three vectors."""
    n, mean, m2 = 0, None, None
    for a in arrays:
        a = np.asarray(a, np.float64)
        if a.shape[0] == 0:
            continue
        nb, mb = a.shape[0], a.mean(0)
        d = a - mb
        m2b = (d * d).sum(0)
        if mean is None:
            n, mean, m2 = nb, mb, m2b
            continue
        delta = mb - mean
        tot = n + nb
        mean = mean + delta * (nb / tot)
        m2 = m2 + m2b + delta * delta * (n * nb / tot)
        n = tot
    if mean is None:
        raise ValueError("joint_stats: no frames")
    var = m2 / n
    # a constant feature: its variance is rounding noise of the mean (sklearn's _is_constant_feature bound), its scale is 1
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2
    scale = np.sqrt(var)
    scale[constant | (scale == 0.0)] = 1.0
    return mean, scale


def write_joint_stats(stats_file, feature_files, reader=None):
    """calc_stats_vc_joint.py:83-127 for the one-to-one recipe: the joint statistics of `/feat_org_lf0` over `feature_files`
    (source + target training lists) into `stats_file` under /mean_feat_org_lf0_jnt and /scale_feat_org_lf0_jnt (float64 [dim])."""
    import hdf5io
    read = reader or hdf5io.read_hdf5
    mean, scale = joint_stats(read(f, "/feat_org_lf0") for f in feature_files)      # one file in memory at a time
    hdf5io.write_hdf5(stats_file, "/mean_feat_org_lf0_jnt", mean)
    hdf5io.write_hdf5(stats_file, "/scale_feat_org_lf0_jnt", scale)
    return mean, scale


def read_joint_stats(stats_file, stdim):
    """train...:296-299: (mean_jnt, std_jnt, mean_jnt_trg, std_jnt_trg) as float32 tensors; the *_trg ones are the [stdim:] part."""
    import hdf5io
    mean = torch.FloatTensor(hdf5io.read_hdf5(stats_file, "/mean_feat_org_lf0_jnt"))
    std = torch.FloatTensor(hdf5io.read_hdf5(stats_file, "/scale_feat_org_lf0_jnt"))
    return mean, std, mean[stdim:].clone(), std[stdim:].clone()


def set_scalers(model_encoder, model_decoder, mean_jnt, std_jnt, stdim):
    """train...:344-347: the encoder's scale_in becomes diag(1/std) with bias -mean/std over all input dimensions, the decoder's
    scale_out diag(std[stdim:]) with bias mean[stdim:] (the mcep part).  mean_jnt / std_jnt: the joint statistics [in_dim]."""
    dev = next(model_encoder.parameters()).device
    mean = torch.as_tensor(np.asarray(mean_jnt), dtype=torch.float32, device=dev)
    std = torch.as_tensor(np.asarray(std_jnt), dtype=torch.float32, device=dev)
    model_encoder.scale_in.weight = torch.nn.Parameter(torch.diag(1.0 / std).unsqueeze(2))
    model_encoder.scale_in.bias = torch.nn.Parameter(-(mean / std))
    model_decoder.scale_out.weight = torch.nn.Parameter(torch.diag(std[stdim:]).unsqueeze(2))
    model_decoder.scale_out.bias = torch.nn.Parameter(mean[stdim:].clone())
    return mean[stdim:], std[stdim:]


def initial_feedback(mean_jnt_trg, std_jnt_trg, batch_size_utt, lat_dim):
    """train...:357-359: (y_in_pp [B,1,2L] zeros, y_in_src = y_in_trg [B,1,out_dim] = (0 - mean) / std of the mcep statistics)."""
    y_pp = torch.zeros(batch_size_utt, 1, 2 * lat_dim, dtype=torch.float32, device=mean_jnt_trg.device)
    y_in = ((0 - mean_jnt_trg) / std_jnt_trg).unsqueeze(0).unsqueeze(0).repeat(batch_size_utt, 1, 1)
    return y_pp, y_in


def adam_state_dict(step):
    """The state of stage4.Stage4Step's optimiser in torch.optim.Adam's state_dict layout (one entry per trainable parameter in
    parameter order, as `optimizer.state_dict()` of train...:373-377 has it), whether the step runs torch.optim.Adam or the flat
    cvae_adam_step."""
    if step.opt is not None:
        return step.opt.state_dict()
    state, o = {}, 0
    step_no = step.step_no                               # (a device read for the fused step: once)
    for i, p in enumerate(step.params):
        n = p.numel()
        if step_no > 0:
            state[i] = {"step": torch.tensor(float(step_no)), "exp_avg": step.exp_avg[o:o + n].view_as(p).detach().cpu().clone(),
                        "exp_avg_sq": step.exp_avg_sq[o:o + n].view_as(p).detach().cpu().clone()}
        o += n
    group = {"lr": step.lr, "betas": tuple(step.betas), "eps": step.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(step.params)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(step, sd):
    """Inverse of adam_state_dict (a checkpoint written by the reference loads the same way, :378-379)."""
    if step.opt is not None:
        step.opt.load_state_dict(sd)
        return
    g = sd["param_groups"][0]
    step.lr, step.betas, step.eps = g["lr"], tuple(g["betas"]), g["eps"]
    o, steps = 0, set()
    step.exp_avg.zero_()
    step.exp_avg_sq.zero_()
    for i, p in enumerate(step.params):
        n = p.numel()
        st = sd["state"].get(i, sd["state"].get(str(i)))
        if st is not None:
            step.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            step.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        o += n
    if len(steps) > 1:
        raise ValueError("parameters with different step counts (%s): not a state of one Adam over all of them" % sorted(steps))
    step.step_no = steps.pop() if steps else 0


def save_checkpoint(checkpoint_dir, model_encoder, model_decoder, optimizer, numpy_random_state, torch_random_state, iterations):
    """train...:152-167: checkpoint-<iterations>.pkl with the keys model_encoder, model_decoder, optimizer, numpy_random_state,
    torch_random_state, iterations.  `optimizer`: a torch optimiser, a stage4.Stage4Step or a ready state_dict.  The models stay
    on their device (the reference moves them to the CPU and back); the saved tensors are CPU tensors like the reference's."""
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    if hasattr(optimizer, "state_dict"):
        opt_sd = optimizer.state_dict()
    elif hasattr(optimizer, "params") and hasattr(optimizer, "grads"):
        opt_sd = adam_state_dict(optimizer)
    else:
        opt_sd = optimizer
    checkpoint = {"model_encoder": cpu(model_encoder.state_dict()), "model_decoder": cpu(model_decoder.state_dict()), "optimizer": opt_sd,
                  "numpy_random_state": numpy_random_state, "torch_random_state": torch_random_state, "iterations": iterations}
    if not os.path.exists(checkpoint_dir):
        os.makedirs(checkpoint_dir)
    path = checkpoint_dir + "/checkpoint-%d.pkl" % iterations
    torch.save(checkpoint, path)
    return path


def resume(path, model_encoder, model_decoder, optimizer=None, restore_rng=False):
    """train...:348-355 (+ :378-379 optimiser, :653-655 random states): returns the iteration count of the checkpoint."""
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    model_encoder.load_state_dict(checkpoint["model_encoder"])
    model_decoder.load_state_dict(checkpoint["model_decoder"])
    for m in (model_encoder, model_decoder):
        if hasattr(m, "weights_changed"):
            m.weights_changed()
    if optimizer is not None:
        if hasattr(optimizer, "load_state_dict"):
            optimizer.load_state_dict(checkpoint["optimizer"])
        else:
            load_adam_state_dict(optimizer, checkpoint["optimizer"])
    if restore_rng:
        np.random.set_state(checkpoint["numpy_random_state"])
        torch.set_rng_state(checkpoint["torch_random_state"])
    return checkpoint["iterations"]
