"""Oracle: RectifiedFlow.model_forward / .sample (reference after/diffusion/model.py).

Test infrastructure -- see oracle/__init__.py."""
import torch

from .denoiser import denoiser_forward

# CFG arrangements (SURVEY.md Appendix A "Samplers' CFG variants")
CFG_API = 0  # model.py:730-759   rows (c,tc)/(-4,tc)/(-4,-4), clamp 0.01
CFG_EXPORT = 1  # export.py:364-394 same rows, clamp 0.1
CFG_MIDI = 2  # export_midi.py:329-358 rows (c,tc)/(c,-4)/(-4,-4), factor g_s/max(g_t,.1)


def model_forward(sd, cfg, x, time, cond, time_cond, guidance_timbre, guidance_structure,
                  drop_value=-4.0, cfg_mode=CFG_API, cache=None, cache_index=0):
    """model.py:721-761."""
    full_time = time.repeat(3, 1, 1)
    full_x = x.repeat(3, 1, 1)
    dc = drop_value * torch.ones_like(cond)
    dt_ = drop_value * torch.ones_like(time_cond)
    if cfg_mode == CFG_MIDI:
        full_cond = torch.cat([cond, cond, dc])
        full_tc = torch.cat([time_cond, dt_, dt_])
    else:
        full_cond = torch.cat([cond, dc, dc])
        full_tc = torch.cat([time_cond, time_cond, dt_])
    dx = denoiser_forward(sd, cfg, full_x, full_time, full_cond, full_tc, cache=cache,
                          cache_index=cache_index)
    dx_full, dx_mid, dx_none = torch.chunk(dx, 3, dim=0)
    total = 0.5 * (guidance_structure + guidance_timbre)
    if cfg_mode == CFG_API:
        factor = guidance_timbre / max(guidance_structure, 0.01)
    elif cfg_mode == CFG_EXPORT:
        factor = guidance_timbre / max(guidance_structure, 0.1)
    else:
        factor = guidance_structure / max(guidance_timbre, 0.1)
    return dx_none + total * (dx_mid + factor * (dx_full - dx_mid) - dx_none)


def sample(sd, cfg, x0, cond, time_cond, nb_steps, guidance_timbre=1.0,
           guidance_structure=1.0, drop_value=-4.0, cfg_mode=CFG_API, return_trajectory=False):
    """model.py:763-785: fixed-step Euler, t = linspace(0,1,N+1)[:-1], dt = 1/N."""
    dt = 1 / nb_steps
    t_values = torch.linspace(0, 1, nb_steps + 1)[:-1]
    x = x0
    traj = []
    for t in t_values:
        tt = t.reshape(1, 1, 1).repeat(x.shape[0], 1, 1).to(x.dtype)
        x = x + model_forward(sd, cfg, x, tt, cond, time_cond, guidance_timbre,
                              guidance_structure, drop_value, cfg_mode) * dt
        if return_trajectory:
            traj.append(x)
    return (x, traj) if return_trajectory else x
