"""Helpers shared by the scenarios that run on the generic path (torch callbacks over the SoA state
views + `mpe_world_step` for the physics)."""
import torch


def dist(a, b):
    """np.sqrt(np.sum(np.square(a.state.p_pos - b.state.p_pos))) per world -> [B]."""
    d = a.state.p_pos - b.state.p_pos
    return torch.sqrt((d * d).sum(dim=1))


def dist2(a, b):
    d = a.state.p_pos - b.state.p_pos
    return (d * d).sum(dim=1)


def is_collision(a, b):
    """strict `dist < size_a + size_b` (simple_tag.py:69-73 and its copies)."""
    return dist(a, b) < (a.size + b.size)


def const(world, values):
    """A per-entity constant vector (a colour) as a [B, w] tensor."""
    t = world.constant(values)
    return t.unsqueeze(0).expand(world.batch_size, t.shape[0])


def assign(obj, name, value):
    """obj.name = value for a per-world tensor that a reset recomputes (a goal-dependent colour, a key): written IN
    PLACE when obj.name already is a private tensor of that shape, so its address survives resets -- a captured HIP
    graph (GraphedStep) keeps reading the live values."""
    old = getattr(obj, name, None)
    if torch.is_tensor(old) and old._base is None and old.shape == value.shape and old.dtype == value.dtype \
            and old.device == value.device:
        old.copy_(value)
    else:
        setattr(obj, name, value.clone() if value._base is not None else value)


def zeros(world, w=None):
    if w is None:
        return torch.zeros(world.batch_size, dtype=torch.float32, device=world.device)
    return torch.zeros((world.batch_size, w), dtype=torch.float32, device=world.device)


def one_hot_rows(world, index, width, scale=1.0):
    """[B, width] with `scale` at column index[b]."""
    out = zeros(world, width)
    out.scatter_(1, index.to(world.device).long().unsqueeze(1), scale)
    return out


def bound(x):
    """simple_tag.py:103-108 / simple_world_comm.py:154-159 boundary penalty, elementwise."""
    far = torch.clamp(torch.exp(2 * x - 2), max=10.0)
    return torch.where(x < 0.9, torch.zeros_like(x), torch.where(x < 1.0, (x - 0.9) * 10, far))
