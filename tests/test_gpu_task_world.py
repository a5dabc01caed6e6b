"""GPU parity tests (run with `-m gpu`) of the TaskEnvironment whose lanes are the agents of ONE world
(`lanes="agents"`, agentmode="interact": shared goal list, one episode; csrc/riab_task_world.hip through
riab_task_world_step / _reset / _goal_vector): against golden vectors produced by the reference's own TaskEnvironment
with several Agent objects (tests/golden/taskworld_*.npz) and, with thousands of agents, against the oracle's TaskWorld.

Tolerances: positions 1e-9 relative (float64 motion); reward totals, terminal flags, the shared list, cache sizes, the
episode table: exact.

STAND-INS: as in tests/test_gpu_task.py — the goldens were generated with interface stubs for gymnasium / pettingzoo
(oracle/ref_shims/); no arithmetic of the recorded runs goes through them."""
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


def _build(riab, g, n_agents):
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, SpatialGoal, Reward
    presets = {v: k for k, v in riab._lib.DECAYS.items()}
    env = SpatialGoalEnvironment(params={"walls": g["user_walls"].tolist()}, possible_goals=[], render_mode="none",
                                 goalcachekws=dict(reset_n_goals=int(g["reset_n_goals"]), reset_orders_goal=True,
                                                   goalorder=str(g["goalorder"]), agentmode="interact"),
                                 episode_terminate_delay=float(g["terminate_delay"]), teleport_on_reset=bool(g["teleport"]),
                                 lanes="agents")
    env.goal_cache.reset_goals = [
        SpatialGoal(env, pos=row[0:2], goal_radius=row[2],
                    reward=Reward(row[3], dt=row[4], expire_clock=float(row[5]), decay=presets[int(row[6])], decay_knobs=[row[7]]))
        for row in g["goal_table"]]
    Ag = riab.Agent(env, {"dt": float(g["dt"]), "n_agents": n_agents})
    return env, Ag


@pytest.mark.parametrize("fname", gu.TASKWORLD_FILES)
def test_task_world_vs_reference(riab, fname):
    """Closed loop, the batch = the agents of one reference TaskEnvironment: same actions and OU normals in; positions,
    every agent's reward total and cache size, the shared goal list, the terminal flag and the episodes out."""
    g = gu.load(fname)
    T, A = g["pos"].shape[:2]
    env, Ag = _build(riab, g, A)
    env.add_agents(Ag)
    for k, s in gu.PRE_SLICES.items():
        setattr(Ag, k, g["state0"][:, s])
    for k in range(T):
        obs, rew, term, trunc, info = env.step(g["action"][k], agent_kwargs={"noise": g["z"][k].T})
        np.testing.assert_allclose(obs.cpu().numpy(), g["pos"][k], rtol=1e-9, atol=1e-12, err_msg=f"step {k}")
        assert np.array_equal(rew.cpu().numpy(), g["reward"][k]), (k, rew.cpu().numpy(), g["reward"][k])
        left = int(g["goals_left"][k])
        term = term.cpu().numpy()
        assert term.all() == (left == 0) and (term == term[0]).all(), k
        if not g["late"][k]:
            assert bool(term[0]) == bool(g["terminal"][k]), k
        lists = env.goal_cache.goal_lists()
        assert (lists == lists[0]).all() and lists[0].tolist() == g["goal_list"][k].tolist(), (k, lists[0], g["goal_list"][k])
        assert np.array_equal(env.goals_left.cpu().numpy(), np.full(A, left)) and len(env.goal_cache) == left
        assert np.array_equal(Ag.reward.active()[2].cpu().numpy(), g["n_rewards"][k]), k
        assert not trunc.any()
        if g["reset"][k]:
            assert term.all()
            env.reset(positions=g["teleport_pos"][k] if bool(g["teleport"]) else None)
    d = env.diagnostics
    assert d["reward_overflow"] == 0 and d["episode_log_overflow"] == 0
    assert d["late_completions"] == int(g["late"].sum())
    ep = env.episodes
    mine = np.array([ep["episode"], ep["start"], ep["end"], ep["duration"]]).T.reshape(-1, 4)
    np.testing.assert_allclose(mine, g["episodes"][:len(mine)], rtol=0, atol=1e-12)
    assert len(mine) == int(g["reset"].sum()) and set(ep["lane"]) <= {-1}
    assert env.episode == int(g["reset"].sum()) + 1
    st = Ag.reward.stats
    assert np.array_equal((st["total_steps_active"] + st["total_steps_inactive"]).cpu().numpy(), np.full(A, T))
    assert np.allclose(st["max"].cpu().numpy(), g["reward"].max(axis=0), rtol=1e-14)


@pytest.mark.parametrize("goalorder,B,delay", [("nonsequential", 1500, 0.03), ("sequential", 1100, 0.0), ("nonsequential", 257, 0.0),
                                               ("sequential", 64, 0.05)])
def test_task_world_many_agents_vs_oracle(riab, goalorder, B, delay):
    """Thousands of agents in one world (several workgroups: the last one to finish walks the shared list), in-kernel noise,
    the goal-seeking policy on the device (get_goal_vector against the shared list), production goal selection and
    teleports; the oracle's TaskWorld replays the bookkeeping from the observed positions: every agent's reward total
    bit for bit, the shared list, terminal flags, episodes."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment, get_goal_vector
    np.random.seed(2)
    T = 220
    goals = [[0.2, 0.25], [0.8, 0.7], [0.5, 0.12], [0.15, 0.85], [0.5, 0.5], [0.53, 0.5], [0.5, 0.54]]
    walls = [[[0.5, 0.25], [0.5, 0.4]]]
    env = SpatialGoalEnvironment(params={"walls": walls}, possible_goal_positions=goals,
                                 goalcachekws=dict(reset_n_goals=5, goalorder=goalorder), goalkws={"goal_radius": 0.06},
                                 episode_terminate_delay=delay, teleport_on_reset=True, seed=11, lanes="agents")
    Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 4, "agent_id0": 1000})
    env.add_agents(Ag)                                       # reset #1
    table = np.array([[g.pos[0], g.pos[1], g.radius] + g.reward.row() for g in env.goal_cache.get_goals()])
    W = orc.TaskWorld(orc.EnvSpec(walls=walls), table, B, goalorder, delay)
    counter = 1
    sel = orc.task_reset_draws(11, counter, 0xFFFFFFFF, len(goals), 5)
    assert env.goal_cache.goal_lists()[0, :5].tolist() == sel
    np.testing.assert_allclose(Ag.pos[7], orc.task_teleport_draw(11, counter, 1000 + 7, orc.EnvSpec()), rtol=1e-15)
    W.reset(0.0, sel)
    t, resets, awards_total, multi = 0.0, 0, 0, 0
    for k in range(T):
        gv = get_goal_vector(Ag)
        if k < 40 and k % 4 == 0 and W.goal_list and orc.GOAL_TIME_ELAPSED not in W.goal_list:
            pos = Ag.pos   # the goal vector itself: head of the list (sequential) / nearest pending goal
            pend = np.array([table[e, 0:2] for e in W.goal_list])
            want = pend[0] - pos if goalorder == "sequential" else \
                (pend[None] - pos[:, None])[np.arange(B), np.argmin(np.linalg.norm(pend[None] - pos[:, None], axis=2), axis=1)]
            np.testing.assert_allclose(gv.cpu().numpy(), want, rtol=1e-12, atol=1e-15)
        if k == 0:   # the reference's list / dict forms of the call
            assert torch.equal(get_goal_vector([Ag])[Ag.name], gv) and torch.equal(get_goal_vector(env.Ags)[Ag.name], gv)
            with pytest.raises(TypeError):
                get_goal_vector(3)
        nrm = torch.linalg.norm(gv, dim=1, keepdim=True)
        act = torch.where(nrm > 0, 0.9 * gv / nrm.clamp_min(1e-300), torch.zeros_like(gv))
        obs, rew, term, trunc, info = env.step(act)
        t = t + 0.01
        n_before = [len(L.rewards) for L in W.lanes]
        totals, wterm = W.step(obs.cpu().numpy(), t)
        got = [len(L.rewards) - n for L, n in zip(W.lanes, n_before)]
        awards_total += sum(max(x, 0) for x in got)
        multi += sum(1 for x in got if x > 0) > 1
        assert np.array_equal(rew.cpu().numpy(), totals), (k, np.nonzero(rew.cpu().numpy() != totals))
        term = term.cpu().numpy()
        assert (term == wterm).all(), k
        lst = env.goal_cache.goal_lists()[0]
        want = [orc.GOAL_TIME_ELAPSED if e == orc.GOAL_TIME_ELAPSED else e for e in W.goal_list]
        assert lst[:len(want)].tolist() == want and (lst[len(want):] == -1).all(), (k, lst, W.goal_list)
        assert np.array_equal(Ag.reward.active()[2].cpu().numpy(), np.array([len(L.rewards) for L in W.lanes])), k
        if wterm:
            env.reset()
            counter += 1
            resets += 1
            sel = orc.task_reset_draws(11, counter, 0xFFFFFFFF, len(goals), 5)
            W.reset(t, sel)
            assert env.goal_cache.goal_lists()[0, :5].tolist() == sel
            np.testing.assert_allclose(Ag.pos[B - 1], orc.task_teleport_draw(11, counter, 1000 + B - 1, orc.EnvSpec()), rtol=1e-15)
    assert resets >= 2 and awards_total >= 10, (resets, awards_total)
    d = env.diagnostics
    assert d["reward_overflow"] == 0 and d["resets"] == resets + 1 and d["late_completions"] == W.late_completions
    ep = env.episodes
    mine = np.array([ep["episode"], ep["start"], ep["end"], ep["duration"]]).T.reshape(-1, 4)
    np.testing.assert_allclose(mine, np.array(W.finished).reshape(-1, 4), rtol=0, atol=0)
    assert env.episode == W.episode


def test_task_world_is_capturable_and_refuses_what_it_cannot_do(riab):
    """The world step inside a captured graph (no workgroup waits for another; the ticket returns to zero) replays like the
    eager calls; masks and agentmode='noninteract' are refused."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    np.random.seed(0)
    with pytest.raises(NotImplementedError):
        SpatialGoalEnvironment(goalcachekws=dict(agentmode="noninteract"), lanes="agents")
    with pytest.raises(ValueError):
        SpatialGoalEnvironment(lanes="herd")

    def world():
        np.random.seed(0)
        env = SpatialGoalEnvironment(possible_goal_positions=[[0.3, 0.3], [0.7, 0.7], [0.5, 0.2]],
                                     goalcachekws=dict(reset_n_goals=3, reset_orders_goal=True), goalkws={"goal_radius": 0.2},
                                     lanes="agents")
        Ag = riab.Agent(env, {"dt": 0.01, "n_agents": 700, "seed": 3})
        env.add_agents(Ag)
        return env, Ag
    env, Ag = world()
    with pytest.raises(ValueError):
        env.reset(mask=np.ones(700, bool))
    L = riab._lib
    env_s, walls = env.device_tables(Ag.state_tensor.device)
    task = env._task_struct()
    st = Ag.state_tensor

    def launch(e, a, t_env):
        s_, _w = e.device_tables(a.state_tensor.device)
        rc = L.lib.riab_task_world_step(s_, e._task_struct(), L.ptr(e.task_state), L.ptr(e._world), L.ptr(a.state_tensor[0]),
                                        L.ptr(a.state_tensor[1]), e._B, float(t_env), L.ptr(e._reward), L.ptr(e._terminal),
                                        L.ptr(e._met), L.ptr(e._cand), L.ptr(e._ticket), L.ptr(e._diag), L.current_stream())
        assert rc == 0
    # eager: three bookkeeping steps at the same positions (the agents do not move: the rewards decay, goals go on the first)
    for i in range(3):
        launch(env, Ag, 0.01 * (i + 1))
    want = (env._reward.clone(), env._world.clone(), env.task_state.clone())
    assert not env._ticket.any().item()
    env2, Ag2 = world()
    assert torch.equal(Ag2.state_tensor[0:2], Ag.state_tensor[0:2])
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        graph = torch.cuda.CUDAGraph()
        snap = (env2._world.clone(), env2.task_state.clone())
        with torch.cuda.graph(graph, stream=stream):
            launch(env2, Ag2, 0.01)
        env2._world.copy_(snap[0])          # (capture does not run; make sure the state is the initial one)
        env2.task_state.copy_(snap[1])
        graph.replay()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    launch(env2, Ag2, 0.02)
    launch(env2, Ag2, 0.03)
    torch.cuda.synchronize()
    assert torch.equal(env2._reward, want[0]) and torch.equal(env2._world, want[1]) and torch.equal(env2.task_state, want[2])
    assert not env2._ticket.any().item() and len(env2.goal_cache) < 3


@pytest.mark.parametrize("goalorder,B,auto_reset", [("nonsequential", 1030, True), ("sequential", 300, True), ("nonsequential", 64, False),
                                                    ("nonsequential", 1024, True), ("sequential", 512, True), ("nonsequential", 256, False),
                                                    ("sequential", 2048, True)])
def test_task_world_step_plan_equals_eager_loop(riab, goalorder, B, auto_reset):
    """env.make_step_plan(auto_reset, scripted_speed) of a one-world task: one native call per step (motion + the world's
    step, its reset when the episode ended — decided on the device — + the next action, the populations) == the eager loop
    `a = speed * goal direction; env.step(a); if terminal: env.reset(); PCs.update()` bit for bit.  Whole 256-agent
    segments: the step is ONE kernel (csrc/riab_step1.hip, TASK & 8: the writer workgroups keep the world's books, the one
    with the last ticket walks the shared list and posts the verdict); other batches: three launches."""
    from ratinabox_amd.contribs.TaskEnvironment import SpatialGoalEnvironment
    T, speed = 160, 11.0 * 0.08

    def build():
        np.random.seed(2)
        env = SpatialGoalEnvironment(params={"walls": [[[0.5, 0.3], [0.5, 0.7]]]},
                                     possible_goal_positions=[[0.2, 0.25], [0.8, 0.7], [0.5, 0.1], [0.1, 0.9], [0.9, 0.1]],
                                     goalcachekws=dict(reset_n_goals=3, goalorder=goalorder), goalkws={"goal_radius": 0.03},
                                     episode_terminate_delay=0.03, teleport_on_reset=True, seed=11, lanes="agents")
        Ag = riab.Agent(env, {"dt": 0.01, "n_agents": B, "seed": 4})
        # (whole segments: a population the one-launch step covers — the default geometry is geodesic in a room with a wall)
        PCs = riab.PlaceCells(Ag, {"n": 40, "wall_geometry": "euclidean"} if B % 256 == 0 else {"n": 40})
        env.add_agents(Ag)
        return env, Ag, PCs

    e1, A1, P1 = build()
    e2, A2, P2 = build()
    plan = e2.make_step_plan(auto_reset=auto_reset, scripted_speed=speed)
    c0 = e1._reset_counter
    resets = 0
    for k in range(T):
        a = e1._goal_vector(speed)
        obs, rew, term, trunc, info = e1.step(a)
        rew, term = rew.clone(), term.clone()
        if auto_reset and bool(term[0].item()):
            e1._reset_counter = c0 + k     # (the plan advances its reset counter every step, reset or not)
            e1.reset()
            resets += 1
        P1.update()
        plan.step(1)
        assert torch.equal(e2.get_reward(), rew), k
        assert torch.equal(e2.terminal, term), k
        assert torch.equal(e1._world, e2._world), k
    assert np.array_equal(A1.pos, A2.pos) and np.array_equal(P1.firingrate, P2.firingrate)
    assert torch.equal(e1.task_state, e2.task_state)
    assert e1.episodes == e2.episodes and len(e1.episodes["episode"]) == resets
    assert resets >= (2 if auto_reset else 0)
    d1, d2 = e1.diagnostics, e2.diagnostics
    assert d1 == d2, (d1, d2)
    assert np.array_equal(A1.history["pos"], A2.history["pos"])
    assert np.array_equal(P1.history["firingrate"], P2.history["firingrate"])
    info = plan.info()
    if B % 256 == 0:
        assert info["fused_steps"] == T and info["launches"] == T + 1, info            # (+ the plan's first scripted action)
        assert dict(A2.diagnostics)["step1_timeouts_recovered"] == 0
    else:
        assert info["launches"] == 3 * T + int(auto_reset) and info["fused_steps"] == 0   # (motion + world step, reset + action / action, rates)
