"""No-GPU checks of the task layer: the oracle's TaskLane restatement against the golden vectors
generated from the reference's TaskEnvironment, and the host-side descriptions (Reward, goals,
GoalCache validation, C-ABI argument checks)."""
import ctypes as C

import numpy as np
import pytest

from tests import golden_util as gu
from oracle import riab_oracle as orc


@pytest.mark.parametrize("fname", gu.TASK_FILES)
def test_task_lane_oracle_vs_reference(fname):
    """Reward totals (bit-exact), terminal flags, goal / reward cache sizes and the episode table of
    every lane of the reference runs, replayed from the recorded positions."""
    g = gu.load(fname)
    env = orc.EnvSpec(walls=g["user_walls"])
    n = int(g["reset_n_goals"])
    late = 0
    for lane in range(g["pos"].shape[0]):
        L = orc.TaskLane(env, g["goal_table"], str(g["goalorder"]), float(g["terminate_delay"]))
        L.reset(0.0, range(n))
        t = 0.0
        for k in range(g["pos"].shape[1]):
            t = t + float(g["dt"])
            total, term = L.step(g["pos"][lane, k], t)
            assert total == g["reward"][lane, k], (lane, k)
            assert len(L.goal_list) == g["goals_left"][lane, k] and len(L.rewards) == g["n_rewards"][lane, k], (lane, k)
            assert term == (g["goals_left"][lane, k] == 0)
            if not g["late"][lane, k]:  # the reference reports the flag of its FIRST check pass
                assert term == bool(g["terminal"][lane, k]), (lane, k)
            if g["reset"][lane, k]:
                L.reset(t, range(n))
        ep = g["episodes"][lane]
        ep = ep[~np.isnan(ep[:, 0])]
        mine = np.array(L.finished).reshape(-1, 4)
        assert len(mine) >= 1 or len(ep) <= 1
        np.testing.assert_allclose(mine, ep[:len(mine)], rtol=0, atol=0)
        late += L.late_completions
    assert late == int(g["late"].sum())


def test_world_list_logic_oracle_vs_reference():
    """The oracle's walk of the SHARED goal list (several agents, agentmode="interact") against the reference's own
    GoalCache.check driven by random who-stands-in-which-goal tables: awards (agent, goal) in order and the list left
    behind, for three consecutive passes (what one step runs)."""
    g = gu.load("taskworld_list_logic.npz")
    for c in range(g["met"].shape[0]):
        na, ng, seq = (int(x) for x in g["dims"][c])
        met = g["met"][c]
        lst = list(range(ng))
        for p in range(g["award_agent"].shape[1]):
            awards = orc.world_check_pass(lst, orc._Met(na, lambda a, e: bool(met[a, e])), bool(seq))
            n = int((g["award_agent"][c, p] >= 0).sum())
            assert [a for a, _ in awards] == g["award_agent"][c, p, :n].tolist(), (c, p)
            assert [e for _, e in awards] == g["award_goal"][c, p, :n].tolist(), (c, p)
            assert lst == g["left_after"][c, p, :len(lst)].tolist() and (g["left_after"][c, p, len(lst):] == -1).all(), (c, p)


@pytest.mark.parametrize("fname", gu.TASKWORLD_FILES)
def test_task_world_oracle_vs_reference(fname):
    """Several agents in ONE reference TaskEnvironment (shared goals, one episode): every agent's reward total
    (bit-exact) and cache size, the shared list, the terminal flag and the episode table, replayed from the
    recorded positions."""
    g = gu.load(fname)
    env = orc.EnvSpec(walls=g["user_walls"])
    n = int(g["reset_n_goals"])
    T, A = g["pos"].shape[:2]
    W = orc.TaskWorld(env, g["goal_table"], A, str(g["goalorder"]), float(g["terminate_delay"]))
    W.reset(0.0, range(n))
    t = 0.0
    for k in range(T):
        t = t + float(g["dt"])
        totals, term = W.step(g["pos"][k], t)
        assert np.array_equal(totals, g["reward"][k]), k
        left = int(g["goals_left"][k])
        assert W.goal_list == g["goal_list"][k, :left].tolist(), k
        assert [len(L.rewards) for L in W.lanes] == g["n_rewards"][k].tolist(), k
        assert term == (left == 0)
        if not g["late"][k]:
            assert term == bool(g["terminal"][k]), k
        if g["reset"][k]:
            W.reset(t, range(n))
    mine = np.array(W.finished).reshape(-1, 4)
    np.testing.assert_allclose(mine, g["episodes"][:len(mine)], rtol=0, atol=0)
    assert len(mine) == int(g["reset"].sum())
    assert W.late_completions == int(g["late"].sum())


@pytest.fixture(scope="module")
def world_logic(tmp_path_factory):
    """csrc/riab_task_world_logic.h — the text the kernel compiles — built for the host with g++ behind a serial driver."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("world_logic") / "world_logic_host.so")
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", "-I", os.path.join(root, "ratinabox_amd", "csrc"),
                    os.path.join(root, "tests", "native", "world_logic_host.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.world_pass_host.restype = C.c_int

    def run(lst, met_bits, pad_elapsed, sequential):
        buf = (C.c_uint8 * 16)(*lst)
        n = C.c_int(len(lst))
        met = (C.c_uint64 * len(met_bits))(*met_bits)
        aa, ae = (C.c_int * 32)(), (C.c_int * 32)()
        k = lib.world_pass_host(buf, C.byref(n), met, len(met_bits), int(pad_elapsed), int(sequential), aa, ae)
        return [(aa[i], ae[i]) for i in range(k)], list(buf[:n.value])
    return run


def test_kernel_list_logic_vs_reference(world_logic):
    """The kernel's list logic (turns only for agents that stand in a goal the turn looks at) against the reference's
    GoalCache.check on the recorded tables, three passes each."""
    g = gu.load("taskworld_list_logic.npz")
    for c in range(g["met"].shape[0]):
        na, ng, seq = (int(x) for x in g["dims"][c])
        bits = [int(sum(1 << e for e in range(ng) if g["met"][c, a, e])) for a in range(na)]
        lst = list(range(ng))
        for p in range(g["award_agent"].shape[1]):
            awards, lst = world_logic(lst, bits, False, seq)
            n = int((g["award_agent"][c, p] >= 0).sum())
            assert awards == list(zip(g["award_agent"][c, p, :n].tolist(), g["award_goal"][c, p, :n].tolist())), (c, p)
            assert lst == g["left_after"][c, p, :len(lst)].tolist() and (g["left_after"][c, p, len(lst):] == -1).all(), (c, p)


def test_kernel_list_logic_vs_oracle_with_the_delay_goal(world_logic):
    """... and against the oracle's walk on random lists that contain the termination-delay goal (pool indices up to 63,
    up to 200 agents, elapsed or not)."""
    rs = np.random.RandomState(5)
    PAD = 0xFE
    for c in range(600):
        na, n = int(rs.randint(1, 200)), int(rs.randint(0, 17))
        lst = rs.choice(64, size=n, replace=False).tolist()
        if n and rs.rand() < 0.5:
            lst[int(rs.randint(n))] = PAD
        dens = rs.choice([0.002, 0.02, 0.3])
        met = rs.random_sample((na, 64)) < dens
        bits = [int(sum(1 << e for e in np.nonzero(met[a])[0])) for a in range(na)]
        elapsed, seq = bool(rs.rand() < 0.5), bool(c % 2)
        ref_list = list(lst)
        ref = orc.world_check_pass(ref_list, orc._Met(na, lambda a, e: elapsed if e == PAD else bool(met[a, e])), seq)
        awards, left = world_logic(lst, bits, elapsed, seq)
        assert awards == ref and left == ref_list, c


def test_reset_draws_are_permutation_prefixes():
    for lane in range(50):
        d = orc.task_reset_draws(seed=7, counter=3, lane_id=lane, n_pool=9, n_select=6)
        assert len(set(d)) == 6 and all(0 <= x < 9 for x in d)
    allsel = np.array([orc.task_reset_draws(1, 1, lane, 5, 1)[0] for lane in range(4000)])
    assert np.all(np.abs(np.bincount(allsel, minlength=5) / 4000 - 0.2) < 0.03)  # uniform over the pool


def test_reward_descriptions():
    from ratinabox_amd.contribs.TaskEnvironment import Reward, reward_default, no_reward_default
    assert reward_default.row() == [1.0, 0.01, 1.0, 1.0, 1.0]
    assert no_reward_default.row() == [0.0, 0.01, 0.1, 3.0, 0.0] == list(orc.PAD_REWARD)
    r = Reward(2, dt=0.02, expire_clock=0.5, decay="exponential")
    assert r.row() == [2.0, 0.02, 0.5, 2.0, 2.0] and r.get_delta() == -2 * np.exp(2)
    assert Reward(1, decay="constant", expire_clock=None).expire_clock == 0.01  # falls back to dt (:780-782)
    with pytest.raises(NotImplementedError):
        Reward(1)  # the reference's default (decay=None) cannot be stepped either (TypeError in its update)
    with pytest.raises(NotImplementedError):
        Reward(1, decay="linear", external_drive=lambda: 1.0)


def test_goal_cache_validation_and_pool():
    from ratinabox_amd.contribs.TaskEnvironment import GoalCache, SpatialGoalEnvironment, SpatialGoal
    with pytest.raises(ValueError):
        GoalCache(None, goalorder="sometimes")
    with pytest.raises(ValueError):
        GoalCache(None, agentmode="compete")
    with pytest.raises(ValueError):
        GoalCache(None, reset_n_goals=0)
    assert GoalCache(None, goalorder="custom").goalorder == "custom"   # (refused by the first goal check, like the reference)
    np.random.seed(3)
    env = SpatialGoalEnvironment(possible_goal_positions="random_4", goalkws={"goal_radius": 0.07})
    pool = env.goal_cache.get_goals()
    assert len(pool) == 4 and all(isinstance(g, SpatialGoal) and g.radius == 0.07 for g in pool)
    assert env.get_goal_positions().shape == (4, 2)
    env2 = SpatialGoalEnvironment(possible_goal_positions=[[0.1, 0.2], [0.3, 0.4]])
    assert env2.goal_cache.get_goals()[0].radius == pytest.approx(0.1)  # min(dx*10, ptp(extent)/10)
    assert env2.goal_cache.get_goals()[1] == [0.3, 0.4]
    with pytest.raises(ValueError):
        SpatialGoalEnvironment(possible_goal_positions="grid_4")
    # the two batchings: every lane its own replica of the task / the lanes as the agents of one world (interact only)
    assert env2.lanes == "replicas" and SpatialGoalEnvironment(lanes="agents").goal_cache.agentmode == "interact"
    with pytest.raises(ValueError):
        SpatialGoalEnvironment(lanes="herd")
    with pytest.raises(NotImplementedError):
        SpatialGoalEnvironment(lanes="agents", goalcachekws=dict(agentmode="noninteract"))
    with pytest.raises(NotImplementedError):
        env2.render()


def test_task_abi_argument_errors():
    from ratinabox_amd import _lib as L
    env, task = L.RiabEnv(), L.RiabTask()
    p = C.c_void_p(64)
    assert L.lib.riab_task_step(None, task, p, p, p, 4, 0.0, p, p, p, None) == -1
    assert L.lib.riab_task_step(env, task, p, p, p, 0, 0.0, p, p, p, None) == -1
    assert L.lib.riab_task_step(env, task, p, None, p, 4, 0.0, p, p, p, None) == -1
    task.n_pool = 65
    assert L.lib.riab_task_step(env, task, p, p, p, 4, 0.0, p, p, p, None) == -3      # RIAB_ETOOBIG
    task.n_pool = 2
    assert L.lib.riab_task_step(env, task, p, p, p, 4, 0.0, p, p, p, None) == -1      # goals pointer missing
    task.goals = 64
    env.periodic = 1
    assert L.lib.riab_task_step(env, task, p, p, p, 4, 0.0, p, p, p, None) == -4      # line of sight needs solid walls
    env.periodic = 0
    task.goalorder = 7
    assert L.lib.riab_task_step(env, task, p, p, p, 4, 0.0, p, p, p, None) == -4
    task.goalorder = 0
    assert L.lib.riab_task_reset(env, task, p, None, 4, 0, 0.0, 16, 1, 0, 0, 0, None, None, None, None, None, None, None, 0, None, p, None) == -3
    assert L.lib.riab_task_reset(env, task, p, None, 4, 0, 0.0, 2, 1, 0, 0, 1, None, None, None, None, None, None, None, 0, None, p, None) == -1
    assert C.sizeof(L.RiabTask) == 8 + 4 + 4 + 8 + 5 * 8 + 8
    # the one-world entry points: the shared state, the scratch and the ticket are required
    task.n_pool, task.goals = 2, 64
    W = L.lib.riab_task_world_step
    assert W(env, task, p, None, p, p, 4, 0.0, p, p, p, p, p, p, None) == -1
    assert W(env, task, p, p, p, p, 4, 0.0, p, p, None, p, p, p, None) == -1
    assert W(env, task, p, p, p, p, 4, 0.0, p, p, p, None, p, p, None) == -1
    assert W(env, task, p, p, p, p, 4, 0.0, p, p, p, p, None, p, None) == -1
    assert W(env, task, p, p, p, p, 1 << 31, 0.0, p, p, p, p, p, p, None) == -3
    assert L.lib.riab_task_world_reset(env, task, p, None, 4, 0, 0.0, 2, 1, 0, 0, 0, None, None, None, None, None, None, None, 0,
                                       None, 0, 0.0, None, None, p, None) == -1
    assert L.lib.riab_task_world_reset(env, task, p, p, 4, 0, 0.0, 16, 1, 0, 0, 0, None, None, None, None, None, None, None, 0,
                                       None, 1, 0.0, None, None, p, None) == -3
    assert L.lib.riab_task_world_reset(env, task, p, p, 4, 0, 0.0, 2, 1, 0, 0, 0, None, None, None, None, None, None, None, 0,
                                       None, 1, 1.0, p, p, p, None) == -1      # a goal vector needs the positions
    assert L.lib.riab_plan_set_task_world(None, p, p, p, p) == -1 and L.lib.riab_plan_discard_ahead(None) == -1
    assert L.lib.riab_task_world_goal_vector(env, task, p, p, p, p, 4, 0.0, None, p, None) == -1
    assert L.TW_ROWS == 24 and L.TW_GOAL_LIST + L.TASK_MAX_GOALS == L.TW_ROWS
