"""GPU parity tests of the batched TaskEnvironment (run with `-m gpu`): `step` / `reset` through the
Python classes and hence riab_task_step / riab_task_reset, against golden vectors produced by the
reference's own TaskEnvironment (one single-agent reference run per lane; tests/golden/task_*.npz)
and against the oracle's TaskLane restatement.

Tolerances: positions 1e-9 relative (float64 motion); reward totals bit-exact (float64 recursions
with the reference's operation order) except the "exponential" decay preset, which goes through the
device exp (<= 1 ulp: 1e-14 relative); terminal flags, goal counts, reward-cache sizes, episode
tables: exact.

STAND-INS: the reference's TaskEnvironment imports gymnasium and pettingzoo, which this image does not have; the
task_*.npz goldens were generated with interface stubs for both (oracle/ref_shims/: `spaces.Box` as a shape holder,
`ParallelEnv` as an empty base class).  The reference uses them to describe spaces and as a marker class only: no
arithmetic of the recorded runs goes through them."""
import numpy as np
import pytest
import torch

from oracle import riab_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def riab():
    assert torch.cuda.is_available(), "these tests need the GPU"
    import ratinabox_amd
    return ratinabox_amd


def _build(riab, g, n_agents=None, teleport=None, **envkw):
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, SpatialGoal, Reward
    presets = {v: k for k, v in riab._lib.DECAYS.items()}
    env = SpatialGoalEnvironment(params={"walls": g["user_walls"].tolist()}, possible_goals=[], render_mode="none",
                                 goalcachekws=dict(reset_n_goals=int(g["reset_n_goals"]), reset_orders_goal=True,
                                                   goalorder=str(g["goalorder"])),
                                 episode_terminate_delay=float(g["terminate_delay"]),
                                 teleport_on_reset=bool(g["teleport"]) if teleport is None else teleport, **envkw)
    env.goal_cache.reset_goals = [
        SpatialGoal(env, pos=row[0:2], goal_radius=row[2],
                    reward=Reward(row[3], dt=row[4], expire_clock=float(row[5]), decay=presets[int(row[6])], decay_knobs=[row[7]]))
        for row in g["goal_table"]]
    B = g["pos"].shape[0] if n_agents is None else n_agents
    Ag = riab.Agent(env, {"dt": float(g["dt"]), "n_agents": B})
    return env, Ag


@pytest.mark.parametrize("fname", gu.TASK_FILES)
def test_task_environment_vs_reference(riab, fname):
    """Closed loop, every lane = one reference run: same actions and OU normals in, positions,
    rewards, terminal flags, cache sizes and episodes out."""
    g = gu.load(fname)
    env, Ag = _build(riab, g)
    B, T = g["pos"].shape[:2]
    env.add_agents(Ag)          # (resets, and teleports when teleport_on_reset: the recorded state is from after it)
    for k, s in gu.PRE_SLICES.items():
        setattr(Ag, k, g["state0"][:, s])
    L = riab._lib
    exact = not np.any(g["goal_table"][:, 6] == L.DECAYS["exponential"])
    for k in range(T):
        obs, rew, term, trunc, info = env.step(g["action"][:, k], agent_kwargs={"noise": g["z"][:, k].T})
        np.testing.assert_allclose(obs.cpu().numpy(), g["pos"][:, k], rtol=1e-9, atol=1e-12, err_msg=f"step {k}")
        r = rew.cpu().numpy()
        if exact:
            assert np.array_equal(r, g["reward"][:, k]), (k, r, g["reward"][:, k])
        else:
            np.testing.assert_allclose(r, g["reward"][:, k], rtol=1e-14, atol=0)
        term = term.cpu().numpy()
        assert np.array_equal(term, g["goals_left"][:, k] == 0), k
        ok = ~g["late"][:, k]
        assert np.array_equal(term[ok], g["terminal"][ok, k]), k  # where the reference's first-pass flag applies
        assert np.array_equal(env.goals_left.cpu().numpy(), g["goals_left"][:, k]), k
        assert np.array_equal(Ag.reward.active()[2].cpu().numpy(), g["n_rewards"][:, k]), k
        assert not trunc.any()
        if g["reset"][:, k].any():
            assert np.array_equal(g["reset"][:, k], term)
            env.reset(mask=term, positions=np.nan_to_num(g["teleport_pos"][:, k]) if bool(g["teleport"]) else None)
    d = env.diagnostics
    assert d["reward_overflow"] == 0 and d["episode_log_overflow"] == 0
    assert d["late_completions"] == int(g["late"].sum())
    ep = env.episodes
    for lane in range(B):
        ref = g["episodes"][lane]
        ref = ref[~np.isnan(ref[:, 3])]
        mine = np.array([[e, s, en, du] for ln, e, s, en, du in zip(ep["lane"], ep["episode"], ep["start"], ep["end"], ep["duration"])
                         if ln == lane]).reshape(-1, 4)
        np.testing.assert_allclose(mine, ref[:len(mine)], rtol=0, atol=1e-12)
        assert len(mine) == int(g["reset"][lane].sum())
    # RewardCache.stats: every step is either active or inactive
    st = Ag.reward.stats
    assert np.array_equal((st["total_steps_active"] + st["total_steps_inactive"]).cpu().numpy(), np.full(B, T))
    assert np.allclose(st["max"].cpu().numpy(), g["reward"].max(axis=1), rtol=1e-14)


def test_task_production_reset_matches_host_philox(riab):
    """Unordered goal selection and teleports of riab_task_reset == the oracle's Philox restatement;
    selections are prefixes of permutations; masked lanes are untouched."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    np.random.seed(0)
    B = 300
    env = SpatialGoalEnvironment(possible_goal_positions="random_9", goalcachekws=dict(reset_n_goals=5), seed=77,
                                 teleport_on_reset=True)
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "agent_id0": 64})
    env.add_agents(Ag)                      # reset #1
    lists = env.goal_cache.goal_lists()
    ospec = orc.EnvSpec()
    for lane in (0, 1, 17, 299):
        assert list(lists[lane, :5]) == orc.task_reset_draws(77, 1, 64 + lane, 9, 5)
        np.testing.assert_allclose(Ag.pos[lane], orc.task_teleport_draw(77, 1, 64 + lane, ospec), rtol=1e-15)
    assert all(len(set(r[:5])) == 5 and (r[5:] == -1).all() for r in lists)
    assert np.all((Ag.pos > 0.05 - 1e-12) & (Ag.pos < 0.95 + 1e-12))       # centre +- 0.45 * scale
    before, pos_before = lists.copy(), Ag.pos.copy()
    mask = np.arange(B) % 3 == 0
    env.reset(mask=mask)                    # reset #2, a third of the lanes
    after = env.goal_cache.goal_lists()
    assert np.array_equal(after[~mask], before[~mask]) and np.array_equal(Ag.pos[~mask], pos_before[~mask])
    assert list(after[3, :5]) == orc.task_reset_draws(77, 2, 64 + 3, 9, 5)
    assert env.diagnostics["resets"] == B + int(mask.sum())
    # no time has passed: the zero-duration episodes were dropped again (TaskEnvironment.py:333-337)
    assert env.episodes["episode"] == [] and np.all(env.episode.cpu().numpy() == 1)


def test_agents_reach_their_goals_closed_loop(riab):
    """The reference's own test (tests/test_taskenv.py::test_agent_can_reach_goal): driven along
    get_goal_vector every lane terminates; here with 2048 lanes, the policy on the device, in-kernel
    noise, per-lane auto-reset, and the oracle's TaskLane replaying a few lanes from the observations."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, get_goal_vector
    np.random.seed(1)
    B, T = 2048, 400
    goals = [[0.2, 0.25], [0.8, 0.7], [0.5, 0.1], [0.15, 0.85]]
    env = SpatialGoalEnvironment(params={"walls": [[[0.5, 0.3], [0.5, 0.7]]]}, possible_goal_positions=goals,
                                 goalcachekws=dict(reset_n_goals=2, goalorder="nonsequential"),
                                 episode_terminate_delay=0.05, seed=5)
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 9})
    env.add_agents(Ag)
    probe = [0, 1, 500, 2047]
    table = np.array([[g.pos[0], g.pos[1], g.radius] + g.reward.row() for g in env.goal_cache.get_goals()])
    lanes = {}
    for b in probe:
        lanes[b] = orc.TaskLane(orc.EnvSpec(walls=[[[0.5, 0.3], [0.5, 0.7]]]), table, "nonsequential", 0.05)
        lanes[b].reset(0.0, env.goal_cache.goal_lists()[b, :2])
    done = torch.zeros(B, dtype=torch.long, device="cuda")
    t = 0.0
    for k in range(T):
        v = get_goal_vector(Ag)
        a = 11.0 * Ag.speed_mean * v / torch.linalg.norm(v, dim=1, keepdim=True)      # NaN where no goal is pending
        obs, rew, term, trunc, info = env.step(a)
        done += term
        t = t + 0.01
        o, r, tm = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        for b in probe:
            total, terminal = lanes[b].step(o[b], t)
            assert total == r[b] and terminal == tm[b], (k, b)
        if term.any():
            env.reset(mask=term)
            sel = env.goal_cache.goal_lists()
            for b in probe:
                if tm[b]:
                    lanes[b].reset(t, sel[b, :2])
    assert (done > 0).float().mean().item() > 0.6            # most lanes finished an episode within 4 s
    ep = env.episodes
    assert len(ep["episode"]) == int(done.sum().item()) and min(ep["duration"]) > 0
    assert env.diagnostics["reward_overflow"] == 0


def test_task_edge_cases(riab):
    from ratinabox_amd.contribs.TaskEnvironment import (SpatialGoalEnvironment, TaskEnvironment, SpatialGoal, Reward,
                                                        TimeElapsedGoal)
    # B = 1: reference-shaped single agent, step1 returns python values
    env = SpatialGoalEnvironment(possible_goal_positions=[[0.5, 0.5]], goalkws={"goal_radius": 2.0})
    Ag = riab.Agent(env, {"dt": 0.01})
    env.add_agents(Ag)
    obs, r, term, trunc, info = env.step1(np.array([0.1, 0.0]))
    assert obs.shape == (2,) and r == 1.0 and term is True and env.episode == 1
    env.reset()
    assert env.episode == 2 and env.episodes["duration"] == [pytest.approx(0.01)]
    # pool smaller than reset_n_goals: warns and uses the whole pool (:1231-1239)
    env = SpatialGoalEnvironment(possible_goal_positions=[[0.5, 0.5]], goalcachekws=dict(reset_n_goals=3))
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": 6})
    with pytest.warns(UserWarning):
        env.add_agents(Ag)
    assert np.array_equal(env.goals_left.cpu().numpy(), np.ones(6))
    # no goals at all: every lane is terminal at once; random motion with actions=None
    env = TaskEnvironment()
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": 5})
    env.add_agents(Ag)
    obs, r, term, trunc, info = env.step()
    assert term.all() and (r == 0).all()
    # unsupported configurations fail loudly
    env = TaskEnvironment(goals=[TimeElapsedGoal(None, reward=Reward(1, decay="none"))])
    Ag = riab.Agent(env, {"dt": 0.01})
    with pytest.raises(NotImplementedError):
        env.add_agents(Ag)
    env = SpatialGoalEnvironment(params={"boundary_conditions": "periodic"}, possible_goal_positions=[[0.5, 0.5]])
    Ag = riab.Agent(env, {"dt": 0.01})
    with pytest.raises(AssertionError):
        env.add_agents(Ag)
    # goalorder="custom": accepted, reset works, the first goal check refuses it — after the agents have moved, as in the
    # reference (GoalCache.check has no branch for it: ValueError("Unknown mode: custom"), TaskEnvironment.py:1137-1138)
    env = SpatialGoalEnvironment(possible_goal_positions=[[0.5, 0.5]], goalcachekws=dict(goalorder="custom"))
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": 4})
    env.add_agents(Ag)
    env.reset()
    p0 = Ag.pos.copy()
    with pytest.raises(ValueError, match="Unknown mode: custom"):
        env.step(np.array([0.1, 0.0]))
    assert not np.array_equal(Ag.pos, p0)
    env = SpatialGoalEnvironment(possible_goal_positions=[[0.5, 0.5]])
    with pytest.raises(NotImplementedError):
        env.add_agents(riab.Agent(env, {"dt": 0.05}))       # agent dt != environment dt
    with pytest.raises(NotImplementedError):
        env.add_agents([riab.Agent(env, {"dt": 0.01}), riab.Agent(env, {"dt": 0.01})])


def test_task_step_plan_equals_eager_loop(riab):
    """env.make_step_plan(auto_reset, scripted_speed): one native call per step == the eager loop
    `a = speed * goal direction; env.step(a); env.reset(mask=terminal); PCs.update()` bit for bit."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, get_goal_vector
    B, T, speed = 257, 150, 11.0 * 0.08

    def build():
        np.random.seed(2)
        env = SpatialGoalEnvironment(params={"walls": [[[0.5, 0.3], [0.5, 0.7]]]},
                                     possible_goal_positions=[[0.2, 0.25], [0.8, 0.7], [0.5, 0.1]],
                                     goalcachekws=dict(reset_n_goals=2, goalorder="sequential"),
                                     episode_terminate_delay=0.03, teleport_on_reset=True, seed=11)
        Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 4})
        PCs = riab.PlaceCells(Ag, {"n": 40})
        env.add_agents(Ag)
        return env, Ag, PCs

    e1, A1, P1 = build()
    e2, A2, P2 = build()
    plan = e2.make_step_plan(auto_reset=True, scripted_speed=speed)
    rews, terms = [], []
    for k in range(T):
        a = e1._goal_vector(speed)
        obs, rew, term, trunc, info = e1.step(a)
        rews.append(rew.clone())
        terms.append(term.clone())
        e1.reset(mask=term)
        P1.update()
        plan.step(1)
        assert torch.equal(e2.get_reward(), rews[-1]) and torch.equal(e2.terminal, terms[-1]), k
    assert np.array_equal(A1.pos, A2.pos) and np.array_equal(P1.firingrate, P2.firingrate)
    assert torch.equal(e1.task_state, e2.task_state)
    assert e1.episodes == e2.episodes and len(e1.episodes["episode"]) > 5
    assert e1.t == e2.t and e1.diagnostics == e2.diagnostics
    assert np.array_equal(A1.history["pos"], A2.history["pos"])
    assert np.array_equal(P1.history["firingrate"], P2.history["firingrate"])
    # teleports are visible in the stored trajectory (agent.history["pos"][-1] = agent.pos) and in the rates
    assert torch.stack(terms).any()
    # user-provided actions through the plan == eager step with the same actions
    plan2 = e2.make_step_plan(auto_reset=False)
    act = torch.randn(B, 2, dtype=torch.float64, device="cuda") * 0.2
    act[3] = float("nan")
    o1, r1, t1, _, _ = e1.step(act)
    P1.update()
    plan2.step(1, drift_velocity=act)
    assert np.array_equal(A1.pos, A2.pos) and torch.equal(e2.get_reward(), r1) and torch.equal(e2.terminal, t1)
    # get_goal_vector: raw vectors point at a pending goal of the lane
    v = get_goal_vector(A1).cpu().numpy()
    lists, pool = e1.goal_cache.goal_lists(), e1.get_goal_positions()
    for b in range(0, B, 37):
        if lists[b, 0] >= 0:
            np.testing.assert_allclose(v[b], pool[lists[b, 0]] - A1.pos[b], rtol=0, atol=1e-15)  # sequential: the head
        else:
            assert np.all(v[b] == 0)


@pytest.mark.parametrize("goalorder", ["nonsequential", "sequential"])
def test_task_maximum_sizes(riab, goalorder):
    """64 goals in the pool, 15 per episode (+ the termination-delay goal = all 16 list slots), large
    radii so that several goals are consumed per step at every list position: the packed 16-byte goal
    list, Philox sampling and reward cache against the oracle's TaskLane for a handful of lanes."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, Reward
    np.random.seed(8)
    B, T = 130, 400
    pool = np.random.rand(64, 2) * 0.9 + 0.05
    env = SpatialGoalEnvironment(possible_goal_positions=pool,
                                 goalkws=dict(goal_radius=0.22, reward=Reward(0.5, dt=0.01, expire_clock=0.04, decay="constant",
                                                                              decay_knobs=[0.3])),
                                 goalcachekws=dict(reset_n_goals=15, goalorder=goalorder),
                                 episode_terminate_delay=0.02, teleport_on_reset=True, seed=21)
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 2})
    env.add_agents(Ag)
    lists = env.goal_cache.goal_lists()
    assert lists.shape == (B, 16) and (lists[:, :15] >= 0).all() and (lists[:, 15] == -1).all()
    assert all(len(set(r[:15])) == 15 for r in lists)
    table = np.array([[g.pos[0], g.pos[1], g.radius] + g.reward.row() for g in env.goal_cache.get_goals()])
    probe = [0, 63, 64, 129]
    lanes = {b: orc.TaskLane(orc.EnvSpec(), table, goalorder, 0.02) for b in probe}
    for b in probe:
        assert list(lists[b, :15]) == orc.task_reset_draws(21, 1, b, 64, 15)
        lanes[b].reset(0.0, lists[b, :15])
    t, max_rw, resets = 0.0, 0, 0
    for k in range(T):
        a = env._goal_vector(30 * 0.08)
        obs, rew, term, _, _ = env.step(a, drift_to_random_strength_ratio=4)
        t = t + 0.01
        o, r, tm = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        nrw = Ag.reward.active()[2].cpu().numpy()
        max_rw = max(max_rw, int(nrw.max()))
        for b in probe:
            total, terminal = lanes[b].step(o[b], t)
            assert total == r[b] and terminal == tm[b] and len(lanes[b].rewards) == nrw[b], (k, b)
        env.reset(mask=term)
        sel = env.goal_cache.goal_lists()
        for b in probe:
            if tm[b]:
                resets += 1
                assert list(sel[b, :15]) == orc.task_reset_draws(21, env._reset_counter, b, 64, 15)
                lanes[b].reset(t, sel[b, :15])
            want = lanes[b].goal_list
            assert list(sel[b, :len(want)]) == [g if g >= 0 else -2 for g in want], (k, b)
    assert env.diagnostics["resets"] > B and max_rw >= 4   # episodes of 15 goals were completed
    assert env.diagnostics["reward_overflow"] == 0


def test_task_plan_with_manual_resets(riab):
    """A task plan without auto-reset, the caller resetting terminal lanes itself between plan steps
    (teleports patch the newest history row the plan wrote) == the eager loop."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    B, T, speed = 48, 120, 12 * 0.08

    def build():
        np.random.seed(6)
        env = SpatialGoalEnvironment(possible_goal_positions=[[0.3, 0.3], [0.7, 0.6]], goalkws=dict(goal_radius=0.15),
                                     goalcachekws=dict(reset_n_goals=1), teleport_on_reset=True, seed=3)
        Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 8})
        PCs = riab.PlaceCells(Ag, {"n": 12})
        env.add_agents(Ag)
        return env, Ag, PCs

    e1, A1, P1 = build()
    e2, A2, P2 = build()
    plan = e2.make_step_plan(auto_reset=False)
    n_term = 0
    for k in range(T):
        a = e1._goal_vector(speed)
        _, r1, t1, _, _ = e1.step(a)
        P1.update()
        e1.reset(mask=t1)
        plan.step(1, drift_velocity=e2._goal_vector(speed))
        assert torch.equal(e2.get_reward(), r1) and torch.equal(e2.terminal, t1), k
        e2.reset(mask=e2.terminal)
        n_term += int(t1.sum().item())
    assert n_term > 10
    assert np.array_equal(A1.pos, A2.pos) and torch.equal(e1.task_state, e2.task_state)
    assert np.array_equal(A1.history["pos"], A2.history["pos"]) and e1.episodes == e2.episodes
    assert np.array_equal(P1.history["firingrate"], P2.history["firingrate"])


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RIAB_TEST_WORLDS", "6"))))
def test_randomised_tasks_vs_oracle(riab, seed):
    """Random tasks (walls, goal pool size / positions / radii, goals per episode, goal order, termination delay,
    reward presets and time constants) in a closed loop with auto-reset, probe lanes replayed through the
    oracle's TaskLane from the observations: reward totals and terminal flags bit for bit."""
    from ratinabox_amd.contribs.TaskEnvironment import (SpatialGoalEnvironment, SpatialGoal, Reward, get_goal_vector)
    rs = np.random.RandomState(4000 + seed)
    np.random.seed(5000 + seed)
    n_walls = int(rs.randint(0, 3))
    walls = [[[float(x), float(rs.uniform(0.05, 0.3))], [float(x), float(rs.uniform(0.6, 0.9))]]
             for x in rs.uniform(0.3, 0.7, n_walls)]
    n_pool = int(rs.randint(2, 9))
    n_sel = int(rs.randint(1, min(4, n_pool) + 1))
    order = str(rs.choice(["nonsequential", "sequential"]))
    delay = float(rs.choice([0.0, 0.03, 0.1]))
    dt = float(rs.choice([0.01, 0.02]))
    B, T = 256, 250
    env = SpatialGoalEnvironment(params={"walls": walls}, possible_goal_positions=[[0.5, 0.5]],
                                 goalcachekws=dict(reset_n_goals=n_sel, goalorder=order),
                                 episode_terminate_delay=delay, teleport_on_reset=bool(rs.randint(0, 2)), seed=int(seed),
                                 dt=dt)
    pool = []
    for _ in range(n_pool):
        preset = str(rs.choice(["linear", "constant", "exponential", "none"]))
        rw = Reward(float(rs.uniform(0.5, 3.0)), dt, expire_clock=float(rs.uniform(0.05, 0.6)), decay=preset,
                    decay_knobs=[float(rs.uniform(0.3, 2.0))])
        pool.append(SpatialGoal(env, pos=rs.uniform(0.08, 0.92, 2), goal_radius=float(rs.uniform(0.04, 0.15)), reward=rw))
    env.goal_cache.reset_goals = pool
    Ag = riab.Agent(env, {"dt": dt, "n_agents": B, "seed": 50 + seed})
    env.add_agents(Ag)
    probe = [0, 7, 100, 255]
    table = np.array([[g.pos[0], g.pos[1], g.radius] + g.reward.row() for g in env.goal_cache.get_goals()])
    oenv = orc.EnvSpec(walls=walls)
    lanes = {}
    for b in probe:
        lanes[b] = orc.TaskLane(oenv, table, order, delay)
        lanes[b].reset(0.0, env.goal_cache.goal_lists()[b, :n_sel])
    t, finished = 0.0, 0
    for k in range(T):
        v = get_goal_vector(Ag)
        a = 9.0 * Ag.speed_mean * v / torch.linalg.norm(v, dim=1, keepdim=True)
        obs, rew, term, trunc, info = env.step(a)
        t = t + dt
        o, r, tm = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        if env.diagnostics["reward_overflow"]:
            # more than RIAB_TASK_MAX_REWARDS (32) rewards alive in one lane: the device cache drops the surplus
            # (counted; the reference's list is unbounded) — nothing left to compare in this world
            return
        for b in probe:
            total, terminal = lanes[b].step(o[b], t)
            # (the exponential preset goes through the device exp: last-ulp differences)
            assert abs(total - r[b]) <= 4e-15 * max(1.0, abs(total)) and terminal == tm[b], \
                (k, b, total, r[b], env.diagnostics)
        if term.any():
            finished += int(term.sum())
            env.reset(mask=term)
            sel = env.goal_cache.goal_lists()
            for b in probe:
                if tm[b]:
                    lanes[b].reset(t, sel[b, :n_sel])
    assert env.diagnostics["episode_log_overflow"] == 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RIAB_TEST_WORLDS", "5"))))
def test_randomised_task_plans_equal_eager_loop(riab, seed):
    """Random tasks (as in test_randomised_tasks_vs_oracle) and batch widths: the step plan — motion + task in one
    launch, auto-reset, scripted action — against the eager `goal vector -> step -> reset(mask) -> update` loop:
    rewards and terminal flags every step, then state, task state, histories and episode tables, bit for bit."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, SpatialGoal, Reward
    rs = np.random.RandomState(9000 + seed)
    n_walls = int(rs.randint(0, 3))
    walls = [[[float(x), float(rs.uniform(0.05, 0.3))], [float(x), float(rs.uniform(0.6, 0.9))]]
             for x in rs.uniform(0.3, 0.7, n_walls)]
    n_pool = int(rs.randint(2, 9))
    n_sel = int(rs.randint(1, min(4, n_pool) + 1))
    order = str(rs.choice(["nonsequential", "sequential"]))
    delay = float(rs.choice([0.0, 0.03, 0.1]))
    dt = float(rs.choice([0.01, 0.02]))
    teleport = bool(rs.randint(0, 2))
    B = int(rs.choice([1, 6, 64, 257, 1024]))
    T = int(rs.randint(40, 160))
    goal_prm = [(rs.uniform(0.08, 0.92, 2), float(rs.uniform(0.04, 0.15)), float(rs.uniform(0.5, 3.0)),
                 float(rs.uniform(0.05, 0.6)), str(rs.choice(["linear", "constant", "exponential", "none"])),
                 float(rs.uniform(0.3, 2.0))) for _ in range(n_pool)]
    speed = float(rs.uniform(4, 12)) * 0.08
    n_cells = int(rs.choice([1, 40]))

    def build():
        np.random.seed(seed)
        env = SpatialGoalEnvironment(params={"walls": walls}, possible_goal_positions=[[0.5, 0.5]],
                                     goalcachekws=dict(reset_n_goals=n_sel, goalorder=order),
                                     episode_terminate_delay=delay, teleport_on_reset=teleport, seed=int(seed), dt=dt)
        env.goal_cache.reset_goals = [
            SpatialGoal(env, pos=p, goal_radius=r, reward=Reward(s0, dt, expire_clock=ex, decay=pre, decay_knobs=[kn]))
            for p, r, s0, ex, pre, kn in goal_prm]
        Ag = riab.Agent(env, {"dt": dt, "n_agents": B, "seed": 60 + seed})
        PCs = riab.PlaceCells(Ag, {"n": n_cells})
        env.add_agents(Ag)
        return env, Ag, PCs

    e1, A1, P1 = build()
    e2, A2, P2 = build()
    plan = e2.make_step_plan(auto_reset=True, scripted_speed=speed)
    for k in range(T):
        obs, rew, term, trunc, info = e1.step(e1._goal_vector(speed))
        e1.reset(mask=term)
        P1.update()
        plan.step(1)
        assert torch.equal(e2.get_reward(), rew) and torch.equal(e2.terminal, term), (k, B, order)
    # (lanes B..B_padded-1 only pad the batch to a multiple of 4: no task, never read back)
    assert torch.equal(A1.state_tensor[:, :B], A2.state_tensor[:, :B]) and torch.equal(e1.task_state, e2.task_state)
    assert e1.episodes == e2.episodes and e1.t == e2.t and e1.diagnostics == e2.diagnostics
    assert np.array_equal(A1.history["pos"], A2.history["pos"])
    assert np.array_equal(P1.history["firingrate"], P2.history["firingrate"])


def _task_world(riab, seed, B, n, teleport, order, delay, pop="place", spikes=False, radius=None):
    n_cells = n
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    np.random.seed(seed)
    env = SpatialGoalEnvironment(params={"walls": [[[0.5, 0.3], [0.5, 0.7]]]},
                                 possible_goal_positions=[[0.2, 0.25], [0.8, 0.7], [0.5, 0.1], [0.3, 0.8]],
                                 goalcachekws=dict(reset_n_goals=2, goalorder=order),
                                 goalkws=dict() if radius is None else dict(goal_radius=radius),
                                 episode_terminate_delay=delay, teleport_on_reset=teleport, seed=seed)
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 4 + seed})
    if pop == "place":
        P = riab.PlaceCells(Ag, {"n": n_cells, "wall_geometry": "euclidean", "save_spikes": spikes, "max_fr": 30 if spikes else 1})
    elif pop == "grid":
        P = riab.GridCells(Ag, {"n": n_cells, "save_spikes": spikes, "max_fr": 30 if spikes else 1})
    else:
        P = riab.HeadDirectionCells(Ag, {"n": n_cells, "save_spikes": False})
    env.add_agents(Ag)
    return env, Ag, P


@pytest.mark.parametrize("case", [
    dict(B=512, n=200, teleport=True, order="sequential", delay=0.03),
    dict(B=256, n=37, teleport=True, order="nonsequential", delay=0.0),
    dict(B=1024, n=64, teleport=False, order="sequential", delay=0.1),
    dict(B=768, n=130, teleport=True, order="nonsequential", delay=0.03, pop="grid"),
    dict(B=256, n=10, teleport=True, order="sequential", delay=0.0, pop="hdc"),
    dict(B=512, n=200, teleport=True, order="sequential", delay=0.03, scripted=False),
    dict(B=512, n=90, teleport=True, order="nonsequential", delay=0.0, auto_reset=False),
    dict(B=256, n=50, teleport=True, order="sequential", delay=0.03, auto_reset=False, scripted=False),
    dict(B=512, n=120, teleport=True, order="nonsequential", delay=0.03, spikes=True),     # (the default: save_spikes)
    # (goals a third of the room wide: many lanes of a wave end an episode in the same step — the positions of a wave's
    # third and later movers travel in their per-agent mail entries, the first two in the verdict)
    dict(B=512, n=64, teleport=True, order="nonsequential", delay=0.0, radius=0.3),
    dict(B=256, n=40, teleport=True, order="sequential", delay=0.03, radius=0.35, pop="grid"),
    dict(B=254, n=48, teleport=True, order="nonsequential", delay=0.03, radius=0.2),   # (padded to 256: two lanes without a task)
    dict(B=256, n=33, teleport=True, order="sequential", delay=0.0, pop="grid", spikes=True, scripted=False),
])
def test_one_launch_task_step_equals_the_two_launch_plan(riab, case):
    """A task plan whose lead population fuses (csrc/riab_step1.hip, TASK modes): Agent.update, the rest of
    TaskEnvironment.step, the reset of the lanes that ended an episode (teleports included: the rates of a teleported
    lane are a function of where it landed), the next scripted action and the population's update() in ONE kernel per
    step — against the same plan with RIAB_OPT_FUSED_STEP = 0 (motion + task kernel, then the rate kernel), which the
    tests above pin against the eager loop, the oracle and the reference goldens.  Everything bit for bit."""
    case = dict(case)
    scripted, auto_reset = case.pop("scripted", True), case.pop("auto_reset", True)
    T, speed = 220, 11.0 * 0.08

    def run(fused):
        old = riab._lib.set_option("fused_step", 1 if fused else 0)
        try:
            env, Ag, P = _task_world(riab, 11, **case)
            plan = env.make_step_plan(auto_reset=auto_reset, scripted_speed=speed if scripted else None)
            rews, terms = [], []
            g = torch.Generator(device="cpu").manual_seed(5)
            for k in range(T):
                if scripted:
                    plan.step(1)
                else:
                    # (the caller's policy: towards the goal, with noise)
                    act = env._goal_vector(speed) + (torch.randn(Ag.n_agents, 2, dtype=torch.float64, generator=g) * 0.1).cuda()
                    plan.step(1, drift_velocity=act)
                rews.append(env.get_reward().clone())
                terms.append(env.terminal.clone())
                if not auto_reset and bool(terms[-1].any()):
                    env.reset(mask=terms[-1])
            torch.cuda.synchronize()
            info = plan.info()
            fr, sp = P.get_history_tensors()
            out = dict(spikes=np.zeros(1) if sp is None else sp.cpu().numpy(), rew=torch.stack(rews).cpu().numpy(), term=torch.stack(terms).cpu().numpy(),
                       state=Ag.state_tensor[:, :Ag.n_agents].cpu().numpy(), ts=env.task_state.cpu().numpy(),
                       traj=Ag.get_history_tensor().cpu().numpy(), fr=fr.cpu().numpy(), last=np.array(P.firingrate),
                       action=plan._actions[:, :Ag.n_agents].cpu().numpy())   # (the coming step's scripted action)
            return out, dict(env.episodes), dict(env.diagnostics), dict(Ag.diagnostics), env.t, info
        finally:
            riab._lib.set_option("fused_step", old)

    a, ep_a, d_a, ad_a, t_a, info_a = run(True)
    b, ep_b, d_b, ad_b, t_b, info_b = run(False)
    # one kernel per closed-loop step (+ the launch that works out the plan's FIRST scripted action)
    assert info_a["fused_steps"] == T and info_a["launches"] == T + int(scripted), info_a
    assert info_b["fused_steps"] == 0
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert ep_a == ep_b and d_a == d_b and ad_a == ad_b and t_a == t_b
    if auto_reset:
        assert a["term"].any() and len(ep_a["episode"]) > 3          # episodes ended, lanes were reset ...
    if case.get("spikes"):
        assert a["spikes"].any()
    if case["teleport"] and auto_reset:
        jumps = np.abs(np.diff(a["traj"][:, 0, :], axis=0)).max()     # ... and teleported (visible in the stored trajectory)
        assert jumps > 0.05


@pytest.mark.parametrize("batch,radius,all_spikes", [(1, None, False), (7, None, False), (1, 0.3, True), (3, 0.35, True)])
def test_one_launch_task_step_with_other_populations_and_batches(riab, batch, radius, all_spikes):
    """The store-bound populations ride in the task step's kernel, the others (boundary vector cells) follow as their own
    kernels on the row the reset patched; `plan.step(batch)` issues `batch` such steps from one native call.  Against the
    two-launch plan, every population, bit for bit.
    `radius`: one goal a third of the room wide per episode, no delay — dozens of lanes of a segment end an episode in the same step, so the rate
    workgroups' second pass (one cell and one moved quad to a lane, seven quads to a round: s1_group_few) runs several
    rounds; `all_spikes`: every population draws spikes (the kernel's groups are then eight cells for every functor)."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    T, speed = 210, 11.0 * 0.08

    def run(fused):
        old = riab._lib.set_option("fused_step", 1 if fused else 0)
        try:
            np.random.seed(21)
            env = SpatialGoalEnvironment(params={"walls": [[[0.4, 0.0], [0.4, 0.5]]]}, possible_goal_positions="random_6",
                                         goalcachekws=dict(reset_n_goals=3 if radius is None else 1, goalorder="nonsequential"),
                                         goalkws=dict() if radius is None else dict(goal_radius=radius),
                                         episode_terminate_delay=0.02 if radius is None else 0.0, teleport_on_reset=True, seed=3)
            Ag = riab.Agent(env, {"dt": 0.01, "n_agents": 512, "seed": 9})
            pops = [riab.HeadDirectionCells(Ag, {"n": 12 if not all_spikes else 41, "save_spikes": all_spikes, "max_fr": 20}),
                    riab.PlaceCells(Ag, {"n": 150, "wall_geometry": "euclidean", "save_spikes": True, "max_fr": 25}),
                    riab.BoundaryVectorCells(Ag, {"n": 20, "save_spikes": False}),
                    riab.GridCells(Ag, {"n": 30, "save_spikes": all_spikes, "max_fr": 15})]
            env.add_agents(Ag)
            plan = env.make_step_plan(neurons=pops, capacity=256, auto_reset=True, scripted_speed=speed)
            for _ in range(T // batch):
                plan.step(batch)
            torch.cuda.synchronize()
            info = plan.info()
            out = dict(state=Ag.state_tensor[:, :512].cpu().numpy(), ts=env.task_state.cpu().numpy(),
                       traj=Ag.get_history_tensor().cpu().numpy(), rew=env.get_reward().cpu().numpy(),
                       term=env.terminal.cpu().numpy())
            for i, P in enumerate(pops):
                fr, sp = P.get_history_tensors()
                out[f"fr{i}"] = fr.cpu().numpy()
                if sp is not None:
                    out[f"sp{i}"] = sp.cpu().numpy()
            return out, dict(env.episodes), dict(env.diagnostics), info
        finally:
            riab._lib.set_option("fused_step", old)

    a, ep_a, d_a, info_a = run(True)
    b, ep_b, d_b, info_b = run(False)
    steps = T // batch * batch
    assert info_a["fused_steps"] == steps and info_a["fused_populations"] == [0, 1, 3], info_a  # (all but the boundary vector cells)
    assert info_a["launches"] == 2 * steps + 1, info_a   # (+ the launch that works out the plan's first scripted action)
    assert info_b["fused_steps"] == 0
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert ep_a == ep_b and d_a == d_b and len(ep_a["episode"]) > 3
    if radius is not None:   # (many lanes reset per step: more than one round of seven quads in a segment's second pass)
        jumps = np.abs(np.diff(a["traj"][:, 0:2, :512], axis=0)).max(1) > 0.05
        assert jumps.sum(1).max() > 16, int(jumps.sum(1).max())
    if all_spikes:
        assert all(a[f"sp{i}"].any() for i in (0, 1, 3))


def test_one_launch_task_step_in_a_captured_graph(riab):
    """The task step's kernels take no decision on the host beyond their scalar arguments and allocate / synchronise
    nothing: four steps captured into a HIP graph and replayed once write what four plain steps write (rows, state, task
    state, episode table, next action), bit for bit."""
    def world():
        env, Ag, P = _task_world(riab, 5, B=256, n=64, teleport=True, order="nonsequential", delay=0.0, radius=0.25)
        plan = env.make_step_plan(capacity=64, auto_reset=True, scripted_speed=0.9)
        plan.step(3)                               # (walls prepared, first action in place, the kernel's attributes asked for)
        torch.cuda.synchronize()
        return env, Ag, P, plan

    def snapshot(env, Ag, P, plan):
        torch.cuda.synchronize()
        return dict(state=Ag.state_tensor.cpu().numpy(), ts=env.task_state.cpu().numpy(), act=plan._actions.cpu().numpy(),
                    rew=env._reward.cpu().numpy(), term=env._terminal.cpu().numpy(), n_ep=env._ep_count.cpu().numpy(),
                    ep=np.sort(env._ep_log.cpu().numpy(), axis=0), diag=env._diag.cpu().numpy(),
                    traj=plan._agent_rows[:7].cpu().numpy(), fr=plan._pop_rows[0][0][:7].cpu().numpy())

    env, Ag, P, plan = world()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rc = riab._lib.lib.riab_plan_step(plan._h, 4, riab._lib.current_stream())
        assert rc == 0
        g.replay()
    torch.cuda.synchronize()
    assert riab._lib.lib.riab_plan_info(plan._h, 0) == 7                 # all seven steps were one-launch steps
    a = snapshot(env, Ag, P, plan)
    env2, Ag2, P2, plan2 = world()
    plan2.step(4)
    b = snapshot(env2, Ag2, P2, plan2)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["n_ep"][0] > 0


@pytest.mark.parametrize("B,fits", [(32768, True), (65536, False)])
def test_one_launch_task_step_only_where_its_grid_is_resident_at_once(riab, B, fits):
    """The task step's workgroups wait for each other: it is used only where all of them fit the chip at once (one per
    compute unit); 65536 agents = 256 writers + 256 others do not, and that plan keeps its two launches."""
    env, Ag, P = _task_world(riab, 2, B=B, n=16, teleport=True, order="nonsequential", delay=0.0)
    plan = env.make_step_plan(capacity=8, auto_reset=True, scripted_speed=0.9)
    plan.step(6)
    torch.cuda.synchronize()
    info = plan.info()
    assert (info["fused_steps"] == 6) == fits, info
    plan.close()                      # (raises if a workgroup of a one-launch step gave up waiting)
    assert np.isfinite(np.asarray(P.firingrate)).all()



@pytest.mark.parametrize("lanes", ["replicas", "agents"])
def test_eager_task_loop_sees_teleports_when_the_update_pair_is_one_launch(riab, lanes):
    """`env.step(a); env.reset(...); PCs.update()` — the reference's order: the population is evaluated AFTER a reset has
    teleported agents (contribs/TaskEnvironment.py:323-330).  With whole 256-agent segments the unchanged loop's
    Agent.update() is the one-launch step, which writes the population's row ahead: a reset in between must make
    PCs.update() recompute it.  Against the same loop with the one-launch step switched off: bit-identical rates."""
    import os
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    B, T, speed = 256, 80, 11.0 * 0.08

    def run(fused):
        if not fused:
            os.environ["RIAB_NO_FUSED_STEP"] = "1"
        try:
            np.random.seed(2)
            env = SpatialGoalEnvironment(possible_goal_positions=[[0.2, 0.25], [0.8, 0.7], [0.5, 0.1]],
                                         goalcachekws=dict(reset_n_goals=2), goalkws={"goal_radius": 0.15},
                                         teleport_on_reset=True, seed=11, lanes=lanes)
            Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 4})
            PCs = riab.PlaceCells(Ag, {"n": 64})
            env.add_agents(Ag)
            resets = fused_steps = 0
            for k in range(T):
                a = env._goal_vector(speed)
                obs, rew, term, trunc, info = env.step(a)
                if lanes == "replicas":
                    env.reset(mask=term)
                    resets += int(term.sum().item())
                elif bool(term[0].item()):
                    env.reset()
                    resets += 1
                PCs.update()
                if Ag._plan is not None and k == T - 1:
                    fused_steps = Ag._plan.info()["fused_steps"]
            return np.array(PCs.history["firingrate"]), np.array(Ag.history["pos"]), resets, fused_steps
        finally:
            os.environ.pop("RIAB_NO_FUSED_STEP", None)

    fr1, pos1, resets1, fused1 = run(True)
    fr0, pos0, resets0, fused0 = run(False)
    assert resets1 == resets0 and resets1 >= 3 and fused1 > 0 and fused0 == 0, (resets1, resets0, fused1, fused0)
    assert np.array_equal(pos1, pos0)
    assert np.array_equal(fr1, fr0), np.nonzero((fr1 != fr0).any(axis=(1, 2)))[0]
