"""Shared test helpers: random reference-encodable states (own generator, numpy only) and trace replay."""
import numpy as np

from oracle.minigrid_oracle import OracleEnvs


def random_states(n, W, H, seed, density=0.4, interior_agent=True):
    """Random walled rooms holding every object kind the hot path distinguishes."""
    rs = np.random.RandomState(seed)
    grid = np.zeros((n, W, H, 3), np.uint8)
    grid[..., 0] = 1
    aux = np.zeros((n, W, H), np.uint8)
    kind = rs.randint(0, 13, size=(n, W, H))
    col = rs.randint(0, 7, size=(n, W, H))
    fill = rs.uniform(size=(n, W, H)) < density
    st = rs.randint(0, 3, size=(n, W, H))
    # kind -> (type, uses_color, state?)
    t = np.select([kind == 0, kind == 1, (kind >= 2) & (kind <= 4), (kind == 5) | (kind == 6), kind == 7, kind == 8,
                   kind == 9, kind == 10, kind >= 11], [2, 3, 4, 5, 6, 7, 8, 8, 9])
    c = np.where(t == 9, 0, np.where(t == 8, 1, col))          # Lava() red, Goal() green
    s = np.where(t == 4, st, 0)
    a = np.where(kind == 10, 0xF1, 0).astype(np.uint8)          # Goal(toggletimes=0): terminal (aux format: include/mgx.h)
    grid[..., 0] = np.where(fill, t, 1)
    grid[..., 1] = np.where(fill, c, 0)
    grid[..., 2] = np.where(fill, s, 0)
    aux[:] = np.where(fill, a, 0)
    # border walls (grey)
    for sl in (np.s_[:, 0, :], np.s_[:, W - 1, :], np.s_[:, :, 0], np.s_[:, :, H - 1]):
        grid[sl] = (2, 5, 0)
        aux[sl] = 0
    agent = np.zeros((n, 3), np.int32)
    agent[:, 0] = rs.randint(1, W - 1, size=n)
    agent[:, 1] = rs.randint(1, H - 1, size=n)
    agent[:, 2] = rs.randint(0, 4, size=n)
    # agent must stand on an overlappable cell (reset() asserts it, minigrid.py:847-848): clear it
    idx = np.arange(n)
    grid[idx, agent[:, 0], agent[:, 1]] = (1, 0, 0)
    aux[idx, agent[:, 0], agent[:, 1]] = 0
    carry = np.tile(np.array([1, 0, 0], np.uint8), (n, 1))
    has = rs.uniform(size=n) < 0.3
    ct = rs.choice([5, 6, 7], size=n)
    carry[has, 0] = ct[has]
    carry[has, 1] = rs.randint(0, 7, size=n)[has]
    steps = rs.randint(0, 5, size=n).astype(np.int32)
    return grid, aux, agent, carry, steps


def random_object_state(grid, seed):
    """Random hidden Goal/Box state for `grid`: aux (toggletimes 0..3, triage_color None/0..2) and Box.contains."""
    rs = np.random.RandomState(seed)
    n, W, H, _ = grid.shape
    tt = rs.randint(0, 4, size=(n, W, H))
    tri = rs.randint(-1, 3, size=(n, W, H))
    is_goal, is_box = grid[..., 0] == 8, grid[..., 0] == 7
    aux = ((((tt - 1) & 15) << 4) | ((tri + 1) << 1)).astype(np.uint8)
    aux = np.where(is_goal, aux | (tt == 0), np.where(is_box, aux, 0)).astype(np.uint8)
    contains = np.zeros((n, W, H, 3), np.uint8)
    contains[..., 0] = 1
    what = rs.randint(0, 3, size=(n, W, H))
    col = rs.randint(0, 7, size=(n, W, H))
    contains[..., 0] = np.where(is_box & (what > 0), 4 + what, 1)
    contains[..., 1] = np.where(is_box & (what > 0), col, 0)
    return aux, contains


def make_oracle(W, H, max_steps, see_through, lava_v1, grid, aux, agent, carry=None, steps=None):
    o = OracleEnvs(W, H, max_steps, see_through, lava_v1)
    o.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
    # the episode start never carries anything / has steps 0 (matches mgx_set_state's snapshot rule)
    return o


def to_np(x):
    return x if isinstance(x, np.ndarray) else x.detach().cpu().numpy()


def onehot(img, nc=7, ns=3):
    """One-hot expansion of (type, color, state) images, restating the reference wrappers' formula:
    OneHotPartialObsWrapper.observation (wrappers.py:226-243): out[.., type] = out[.., 11+color] = out[.., 18+state] = 1
    FullyObsOneHotWrapper.observation (wrappers.py:391-415): class (11) | color (7, or 0 with drop_color) | state (4)."""
    img = np.asarray(img)
    out = np.zeros(img.shape[:-1] + (11 + nc + ns,), np.uint8)
    np.put_along_axis(out, img[..., 0:1].astype(np.int64), 1, axis=-1)
    if nc:
        np.put_along_axis(out, 11 + img[..., 1:2].astype(np.int64), 1, axis=-1)
    np.put_along_axis(out, 11 + nc + img[..., 2:3].astype(np.int64), 1, axis=-1)
    return out
