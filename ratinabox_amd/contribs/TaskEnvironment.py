"""`TaskEnvironment` — the closed-loop caller of the hot path, batched on device
(reference ratinabox/contribs/TaskEnvironment.py).

The reference wraps ONE python `Agent` per pettingzoo agent and, every `step(actions)`, calls
`Agent.update(drift_velocity=action)` then walks python lists of `Goal` / `Reward` objects
(TaskEnvironment.py:361-453).  Here one batched `Agent(n_agents=B)` is added to the environment and
every agent of the batch ("lane") is an independent single-agent replica of the reference's task:
same goal pool, same clock, its own goal list, reward cache and episode counter.  `step(actions)` is
`riab_agent_step` (the motion kernel, `drift_velocity = actions`) followed by ONE launch of
`riab_task_step` (csrc/riab_task.hip) that does the reward decay, goal checks (line-of-sight
distance to `SpatialGoal`s, `TimeElapsedGoal` termination delay) and reward totals of all lanes with
the reference's list semantics; `reset(mask=...)` is `riab_task_reset` for the selected lanes.
Observations, rewards and terminal flags come back as DEVICE tensors with a leading lane axis (a
policy network can consume them and produce the next actions without leaving the GPU); nothing in
`step` synchronises.

Same names and constructor arguments as the reference: `TaskEnvironment`, `SpatialGoalEnvironment`,
`Reward`, `Goal`, `SpatialGoal`, `TimeElapsedGoal`, `GoalCache`, `RewardCache`, `get_goal_vector`,
`reward_default`, `no_reward_default`.  Outside the accelerated path (raise NotImplementedError):
rendering, custom python decay / external-drive functions on rewards, user `TimeElapsedGoal`s in
the pool, several Agent objects (put the agents in one batched Agent).  `goalorder="custom"` behaves as in the
reference: accepted by the constructor, `ValueError("Unknown mode: custom")` from the first step's goal check.

Two batchings (`lanes=`).  "replicas" (default): as above — with one agent per replica `agentmode` "interact" and
"noninteract" coincide.  "agents": the lanes are the agents of ONE world, the reference's multi-agent TaskEnvironment
with `agentmode="interact"` (its default, TaskEnvironment.py:1030): one episode, one SHARED goal list — a goal is
consumed for everybody by the first agent, in agent order, found inside it (GoalCache.check / pop, :1076-1172) — and a
reward cache per agent (`riab_task_world_step` / `_reset`, csrc/riab_task_world.hip)."""
import copy
import random
import warnings

import numpy as np
import torch

from .. import _lib
from ..Agent import Agent
from ..Environment import Environment

_L = _lib


class Box:
    """Descriptor of an action / observation range (stand-in for gymnasium.spaces.Box, which the
    reference only uses to DESCRIBE its spaces, TaskEnvironment.py:197-203)."""

    def __init__(self, low, high, shape=None, dtype=float):
        self.low = np.asarray(low, dtype=float)
        self.high = np.asarray(high, dtype=float)
        self.shape = tuple(shape) if shape is not None else np.shape(self.low)
        self.dtype = dtype


# ------------------------------------------------------------------------------------------------
# rewards and goals: host-side descriptions (the per-lane dynamics run in riab_task_step)
# ------------------------------------------------------------------------------------------------
class Reward:
    """Description of the reward a goal hands out (reference Reward, TaskEnvironment.py:717-832):
    initial value, its own `dt`, time to expiry and one of the decay presets
    ("constant", "linear", "exponential", "none" with `decay_knobs`)."""

    decay_knobs_preset = {"linear": [1], "constant": [1], "exponential": [2], "none": [0]}

    def __init__(self, init_state=1, dt=0.01, expire_clock=None, decay=None, decay_knobs=[], external_drive=None,
                 external_drive_strength=1, name=None):
        if callable(init_state):
            init_state = init_state()
        self.state = init_state
        self.dt = dt
        self.expire_clock = expire_clock if isinstance(expire_clock, (int, float)) else dt
        if not isinstance(decay, str) or decay not in _L.DECAYS:
            raise NotImplementedError("on the accelerated path `decay` must be one of the presets "
                                      f"{list(_L.DECAYS)} (python decay functions cannot run in the kernel)")
        if external_drive is not None:
            raise NotImplementedError("external_drive functions are outside the accelerated path")
        self.preset = decay
        self.decay_knobs = list(decay_knobs) or list(self.decay_knobs_preset[decay])
        self.external_drive = None
        self.external_drive_strength = external_drive_strength
        self.name = name if name is not None else self.__class__.__name__ + " " + str(hash(self))[:5]
        self.goal = None

    def row(self):
        """(init_state, dt, expire_clock, preset code, knob): the five numbers the kernel needs."""
        return [float(self.state), float(self.dt), float(self.expire_clock), float(_L.DECAYS[self.preset]),
                float(self.decay_knobs[0])]

    def get_delta(self, state=None):
        """d(reward)/dt (TaskEnvironment.py:823-832), host-side convenience."""
        state = self.state if state is None else state
        a = self.decay_knobs[0]
        return -{"constant": a, "linear": a * state, "exponential": a * np.exp(state), "none": 0}[self.preset]


reward_default = Reward(1, 0.01, expire_clock=1, decay="linear")
no_reward_default = Reward(0, 0.01, expire_clock=0.1, decay="none")  # what the termination-delay goal gives


class Goal:
    """Base class of goals (TaskEnvironment.py:954-998)."""

    def __init__(self, env=None, reward=reward_default, name=None, **kws):
        self.env = env
        self.reward = reward
        self.reward.goal = self
        self.name = name if name is not None else self.__class__.__name__ + " " + str(hash(random.random()))[:5]

    def check(self, agents=None):
        raise NotImplementedError("check() must be implemented")

    def __call__(self):
        pass


class TimeElapsedGoal(Goal):
    """Satisfied `wait_time` seconds after its creation (TaskEnvironment.py:1262-1278).  The
    environment creates one per lane for `episode_terminate_delay`; user instances in the goal
    pool are not supported on the accelerated path."""

    def __init__(self, *args, wait_time=1, verbose=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.start_time = self.env.t if self.env is not None else 0.0
        self.wait_time = wait_time


class SpatialGoal(Goal):
    """Reach a position: satisfied when the line-of-sight distance from the agent to `pos` is below
    `goal_radius` (TaskEnvironment.py:1281-1373)."""

    def __init__(self, *positionals, pos=None, goal_radius=None, **kws):
        super().__init__(*positionals, **kws)
        if pos is not None:
            self.pos = np.array(pos, dtype=float)
        else:
            self.pos = np.random.rand(int(len(self.env.extent) / 2))
        self.radius = (np.min((self.env.dx * 10, np.ptp(self.env.extent) / 10)) if goal_radius is None else goal_radius)

    def check(self, agents=None):
        """Lanes currently inside the goal radius with a clear line of sight (boolean device tensor)."""
        return self.env._lanes_in_goal(self)

    def __eq__(self, other):
        if isinstance(other, SpatialGoal):
            return bool(np.all(self.pos == other.pos))
        if isinstance(other, (np.ndarray, list)):
            return bool(np.all(self.pos == np.array(other)))
        return NotImplemented

    __hash__ = object.__hash__

    def __call__(self):
        return np.array(self.pos)


class GoalCache:
    """Configuration of how goals are drawn and consumed (TaskEnvironment.py:1001-1259); the
    per-lane goal lists live on the device (`TaskEnvironment.task_state`)."""

    def __init__(self, env, goalorder="nonsequential", agentmode="interact", reset_goals=[], reset_n_goals=1,
                 reset_orders_goal=False, verbose=False, **kws):
        self.env = env
        # ("custom" is accepted here and refused by the first goal check, as in the reference: its GoalCache.check has
        # no branch for it and raises ValueError("Unknown mode: custom"), TaskEnvironment.py:1094-1138)
        if goalorder not in ("sequential", "nonsequential", "custom"):
            raise ValueError("goalorder must be 'sequential', 'nonsequential', or 'custom'")
        if agentmode not in ("interact", "noninteract"):
            raise ValueError("agentmode must be 'interact' or 'noninteract'")
        if reset_n_goals <= 0:
            raise ValueError("reset_n_goals must be > 0")
        self.goalorder = goalorder
        self.agentmode = agentmode
        self.reset_goals = reset_goals
        self.reset_n_goals = reset_n_goals
        self.reset_orders_goal = reset_orders_goal
        self.verbose = verbose

    def get_goals(self):
        """The pool the lanes' goal lists index into."""
        return tuple(self.reset_goals)

    @property
    def goals_left(self):
        """Pending goals per lane (device int tensor)."""
        return self.env.goals_left

    def goal_lists(self):
        """(B, <=16) int array of pool indices per lane, -1 padded (-2: the termination-delay goal)."""
        return self.env._goal_lists()

    def __len__(self):
        """Pending goals summed over the lanes — of the one shared list when the lanes are the agents of one world, like
        the reference's `len(goal_cache)` (synchronises)."""
        if self.env._shared:
            return int(self.env.goals_left[0].item())
        return int(self.env.goals_left.sum().item())


class RewardCache:
    """Per-lane view of the active rewards (reference RewardCache, TaskEnvironment.py:876-945)."""

    def __init__(self, env, default_reward_level=0, verbose=False):
        self.env = env
        self.default_reward_level = default_reward_level
        self.verbose = verbose

    def get_total(self):
        """Current reward of every lane (device float64 tensor)."""
        return self.env.get_reward()

    def active(self):
        """(states [R,B], expire clocks [R,B], count [B]) of the rewards in the lanes' caches."""
        e = self.env
        ts, B = e.task_state, e._B
        return (ts[_L.TS_RW_STATE:_L.TS_RW_STATE + _L.TASK_MAX_REWARDS, :B],
                ts[_L.TS_RW_EXPIRE:_L.TS_RW_EXPIRE + _L.TASK_MAX_REWARDS, :B], ts[_L.TS_N_REWARDS, :B].long())

    @property
    def stats(self):
        ts, B = self.env.task_state, self.env._B
        return {"total_steps_active": ts[_L.TS_STEPS_ACTIVE, :B].long(), "total_steps_inactive": ts[_L.TS_STEPS_INACTIVE, :B].long(),
                "max": ts[_L.TS_R_MAX, :B], "min": ts[_L.TS_R_MIN, :B]}


# ------------------------------------------------------------------------------------------------
class TaskEnvironment(Environment):
    """Environment with task structure (reference TaskEnvironment.py:30-145), one replica per lane — or, with
    `lanes="agents"`, the batch as the agents of ONE world (the reference with several Agents, agentmode="interact").

    One deliberate deviation, in both forms, counted in `diagnostics["late_completions"]`: the reference's `step()` calls
    `_is_terminal_state()` a third time (:438, while it prunes `self.agents`) and throws that call's verdict away — a goal
    the agents complete in THAT pass (overlapping goals, nonsequential order) empties the list without the step
    reporting `terminal`, and the next step appends the `episode_terminate_delay` padding goal.  Here the flag handed
    back (`terminal`, `RIAB_TW_TERMINAL`) is the list's state after that third pass: such a step reports the end of the
    episode at once, and a device-decided reset (`auto_reset`) fires in the same step, without the padding goal.  Reward
    totals are unaffected (the third pass's awards are appended either way)."""

    default_params = {}
    metadata = {"render_modes": ["none"], "name": "TaskEnvironment-RiaB"}

    def __init__(self, *pos, dt=0.01, render_mode="none", render_every=None, render_every_framestep=2,
                 teleport_on_reset=False, save_expired_rewards=False, goals=[], goalcachekws=dict(),
                 rewardcachekws=dict(), episode_terminate_delay=0, verbose=False, seed=0,
                 episode_log_capacity=1 << 20, lanes="replicas", **kws):
        super().__init__(*pos, **kws)
        if lanes not in ("replicas", "agents"):
            raise ValueError("lanes must be 'replicas' (every lane its own copy of the task) or 'agents' (one shared world)")
        self.lanes = lanes
        self._shared = lanes == "agents"
        self.Ags = {}
        self.goal_cache = GoalCache(self, **goalcachekws)
        if self._shared and self.goal_cache.agentmode != "interact":
            raise NotImplementedError("lanes='agents' shares the goals between the agents: agentmode='interact' "
                                      "(the reference's default); use lanes='replicas' for independent goal lists")
        self.goal_cache.reset_goals = goals if isinstance(goals, list) else [goals]
        self.t = 0
        self.dt = dt
        self.history = {"t": []}
        self.verbose = verbose
        self.render_mode = render_mode
        self.teleport_on_reset = teleport_on_reset
        self.save_expired_rewards = save_expired_rewards
        self.observation_spaces = {}
        self.action_spaces = {}
        self.agent_names = []
        self.agents = []
        self.infos = {}
        self.observation_lambda = {}
        self.episode_terminate_delay = episode_terminate_delay
        self.rewardcachekws = rewardcachekws
        self.reward_caches = {}
        self._task_seed = int(seed)
        self._reset_counter = 0
        self._ep_cap = int(episode_log_capacity)
        self._agent = None
        self._pool_rows = None

    # ---- spaces ---------------------------------------------------------------------------------
    def observation_space(self, agent_name):
        return self.observation_spaces[agent_name]

    def action_space(self, agent_name):
        return self.action_spaces[agent_name]

    # ---- agents ---------------------------------------------------------------------------------
    def add_agents(self, agents, names=None, maxvel=50.0, **kws):
        """Attach THE batched Agent (reference add_agents, TaskEnvironment.py:153-218).  Every one of
        its `n_agents` becomes a lane with its own goals, rewards and episodes."""
        if isinstance(agents, (list, tuple)) and len(agents) == 1:
            agents = agents[0]
        if isinstance(agents, dict) and len(agents) == 1:
            names, agents = list(agents.keys()), list(agents.values())[0]
        if not isinstance(agents, Agent):
            if isinstance(agents, (list, tuple, dict)):
                raise NotImplementedError("one batched Agent per TaskEnvironment: use Agent(params={'n_agents': B})")
            raise TypeError("agents must be a list of agents or an agent type")
        if self._agent is not None:
            raise NotImplementedError("one batched Agent per TaskEnvironment: use Agent(params={'n_agents': B})")
        agent = agents
        if agent.dt != self.dt:
            raise NotImplementedError("Does not yet support agents with different dt from envrionment")
        assert self.boundary_conditions == "solid", \
            "line of sight geometry not available for periodic boundary conditions"  # Environment.py:710-713
        name = names[0] if names else "agent_0"
        self._agent = agent
        self.Ags[name] = agent
        self.agent_names.append(name)
        agent.name = name
        D = int(self.dimensionality[0])
        self.action_spaces[name] = Box(low=-maxvel, high=maxvel, shape=(D,))
        ext = [self.extent[i:i + 2] for i in np.arange(0, len(self.extent), 2)]
        lows, highs = np.array(list(zip(*ext)), dtype=float)
        self.observation_spaces[name] = Box(low=lows, high=highs, dtype=float)
        self.observation_lambda[name] = lambda ag: ag.state_tensor[0:2, :ag.n_agents].t()
        cache = RewardCache(self, **self.rewardcachekws)
        self.reward_caches[name] = cache
        agent.reward = cache
        agent.t = self.t
        self.infos[name] = {}
        # device state of the lanes
        dev = agent.state_tensor.device
        self._B = agent.n_agents  # (the agent tensors are padded to a multiple of 4 lanes; the task's are not)
        ts = torch.zeros((_L.TS_ROWS, self._B), dtype=torch.float64, device=dev)
        ts[_L.TS_R_MAX] = -float("inf")
        ts[_L.TS_R_MIN] = float("inf")
        self.task_state = ts
        self._reward = torch.zeros(self._B, dtype=torch.float64, device=dev)
        self._terminal = torch.zeros(self._B, dtype=torch.uint8, device=dev)
        self._truncated = torch.zeros(self._B, dtype=torch.bool, device=dev)
        self._diag = torch.zeros(4, dtype=torch.int32, device=dev)
        self._ep_log = torch.zeros((self._ep_cap, 5), dtype=torch.float64, device=dev)
        self._ep_count = torch.zeros(1, dtype=torch.int32, device=dev)
        if self._shared:   # the world's own state, the lanes' "stands inside" masks, the last-workgroup ticket
            self._world = torch.zeros(_L.TW_ROWS, dtype=torch.float64, device=dev)
            self._met = torch.zeros(self._B, dtype=torch.int64, device=dev)
            self._cand = torch.zeros(self._B, dtype=torch.int32, device=dev)
            self._ticket = torch.zeros(2, dtype=torch.int32, device=dev)
        self.reset()

    def remove_agents(self, agents=None):
        raise NotImplementedError("lanes are fixed by the batched Agent; build a new TaskEnvironment instead")

    def _agentnames(self, agents=None):
        return list(self.agent_names)

    def _dict(self, V):
        return {name: V for name in self.agent_names}

    # ---- goal pool -> device ------------------------------------------------------------------------
    def _task_struct(self, refresh=False):
        """RiabTask for the kernels.  The goal pool is read at reset() time (refresh=True); between
        resets the cached table is used (edits to goals take effect at the next reset)."""
        if not refresh and getattr(self, "_task_cached", None) is not None:
            return self._task_cached
        pool = self.goal_cache.reset_goals
        if len(pool) > _L.TASK_MAX_POOL:
            raise ValueError(f"at most {_L.TASK_MAX_POOL} goals in the pool, got {len(pool)}")
        rows = []
        for g in pool:
            if not isinstance(g, SpatialGoal):
                raise NotImplementedError(f"{type(g).__name__} in the goal pool: only SpatialGoal is accelerated")
            rows.append([float(g.pos[0]), float(g.pos[1]), float(g.radius)] + g.reward.row())
        rows = np.array(rows, dtype=np.float64).reshape(-1, 8)
        if self._pool_rows is None or rows.shape != self._pool_rows.shape or not np.array_equal(rows, self._pool_rows):
            self._pool_rows = rows
            self._pool_dev = torch.from_numpy(rows if len(rows) else np.zeros((1, 8))).to(self._agent.state_tensor.device)
            self._pool_changed = True
        t = _L.RiabTask()
        t.goals = self._pool_dev.data_ptr()
        t.n_pool = len(rows)
        t.goalorder = _L.GOALORDERS.get(self.goal_cache.goalorder, 0)   # ("custom": step() raises before any check)
        t.terminate_delay = float(self.episode_terminate_delay or 0.0)
        for i, v in enumerate(no_reward_default.row()):
            t.pad_reward[i] = v
        t.default_reward_level = float(self.reward_caches[self.agent_names[0]].default_reward_level)
        self._task_cached = t
        return t

    # ---- episode control --------------------------------------------------------------------------
    def seed(self, seed=None):
        """Seed numpy's global generator (like the reference) and the Philox key of the lane resets."""
        np.random.seed(seed)
        if seed is not None:
            self._task_seed = int(seed)

    def reset(self, seed=None, episode_meta_info=False, options=None, mask=None, positions=None, goal_selection=None):
        """Start a new episode (reference reset, TaskEnvironment.py:307-351) in the lanes selected by
        `mask` (boolean (B,), host or device; None = every lane): close the running episode, refill
        the goal list with `goal_cache.reset_n_goals` goals of the pool (the first ones when
        `reset_orders_goal`, else a uniform sample without replacement per lane; or the pool indices
        given per lane in `goal_selection (B, n)`), teleport when `teleport_on_reset` (or to
        `positions (B, 2)` when given).  Active rewards persist across episodes, as in the reference.
        Returns `(observation, infos)`."""
        if seed is not None:
            self.seed(seed)
        if self._agent is None:
            self.agents = copy.copy(self.agent_names)
            return self.get_observation(), self.infos
        ag = self._agent
        ag._sync_plan()  # (a step plan may have advanced the newest history row since the last host-side look)
        dev = ag.state_tensor.device
        task = self._task_struct(refresh=True)
        if getattr(self, "_pool_changed", False):
            if mask is not None and self._reset_counter > 0:
                raise ValueError("the goal pool changed: reset every lane (mask=None) so that no lane keeps stale goals")
            self._pool_changed = False
        n_pool = task.n_pool
        n_sel = self.goal_cache.reset_n_goals
        if n_pool < n_sel:
            warnings.warn(f"Not enough goals to replenish n={n_sel} \nlen(goals)={n_pool}")
            n_sel = n_pool
        if n_sel > _L.TASK_MAX_GOALS - 1:
            raise ValueError(f"at most {_L.TASK_MAX_GOALS - 1} goals per episode, got {n_sel}")
        m = None
        if self._shared and mask is not None:
            raise ValueError("lanes='agents': the agents share one episode, reset() takes no mask")
        if mask is not None:
            m = torch.as_tensor(mask, device=dev).to(torch.uint8).reshape(-1)
            assert m.shape[0] == self._B, "mask must have one entry per agent of the batch"
            m = m.contiguous()
        newpos = None
        teleport = bool(self.teleport_on_reset)
        if positions is not None:
            p = torch.as_tensor(np.asarray(positions, dtype=np.float64) if not torch.is_tensor(positions) else positions)
            newpos = p.to(dev, torch.float64).reshape(self._B, 2).t().contiguous()
            teleport = True
        ordered = bool(self.goal_cache.reset_orders_goal) or goal_selection is not None
        env_s, walls = self.device_tables(dev)
        hist_row = getattr(ag, "_last_row", None)  # newest fp32 row [8, B_padded]: what the rate kernels read
        st = ag.state_tensor
        self._reset_counter += 1
        # (a mask on the host that selects no lane moves nobody: the rows written ahead stay valid)
        nobody = mask is not None and not (torch.is_tensor(mask) and mask.is_cuda) and not bool(np.any(np.asarray(mask)))
        if teleport and not nobody and ag._plan is not None and hasattr(ag._plan, "discard_ahead"):
            # the unchanged per-step loop served natively: this step's Agent.update() may have written its population's row
            # ahead, at the positions the agents are now teleported away from — that population's update() recomputes it
            ag._plan.discard_ahead()
        if self._shared:
            rc = _L.lib.riab_task_world_reset(env_s, task, _L.ptr(self.task_state), _L.ptr(self._world), self._B,
                                              int(ag.agent_id0), float(self.t), int(n_sel), int(ordered), self._task_seed,
                                              self._reset_counter, int(teleport),
                                              _L.ptr(None if newpos is None else newpos[0]),
                                              _L.ptr(None if newpos is None else newpos[1]), _L.ptr(st[0]), _L.ptr(st[1]),
                                              _L.ptr(None if hist_row is None else hist_row[0]),
                                              _L.ptr(None if hist_row is None else hist_row[1]),
                                              _L.ptr(self._ep_log), self._ep_cap, _L.ptr(self._ep_count), 0, 0.0, None, None,
                                              _L.ptr(self._diag),
                                              _L.current_stream())
            _L.check(rc, "riab_task_world_reset")
            self._keep = (newpos, walls, task, self._pool_dev)
            if goal_selection is not None:   # one list for everybody: (n,) pool indices
                sel = torch.as_tensor(np.asarray(goal_selection), device=dev).to(torch.float64).reshape(-1)
                assert sel.shape[0] == n_sel, f"goal_selection must name {n_sel} goals"
                self._world[_L.TW_GOAL_LIST:_L.TW_GOAL_LIST + n_sel].copy_(sel)
            self.agents = copy.copy(self.agent_names)
            return self.get_observation(), self.infos
        rc = _L.lib.riab_task_reset(env_s, task, _L.ptr(self.task_state), _L.ptr(m), self._B, int(ag.agent_id0),
                                    float(self.t), int(n_sel), int(ordered), self._task_seed, self._reset_counter,
                                    int(teleport), _L.ptr(None if newpos is None else newpos[0]),
                                    _L.ptr(None if newpos is None else newpos[1]), _L.ptr(st[0]), _L.ptr(st[1]),
                                    _L.ptr(None if hist_row is None else hist_row[0]),
                                    _L.ptr(None if hist_row is None else hist_row[1]),
                                    _L.ptr(self._ep_log), self._ep_cap, _L.ptr(self._ep_count), _L.ptr(self._diag),
                                    _L.current_stream())
        _L.check(rc, "riab_task_reset")
        self._keep = (m, newpos, walls, task, self._pool_dev)
        if goal_selection is not None:
            sel = torch.as_tensor(np.asarray(goal_selection), device=dev).to(torch.float64).reshape(self._B, -1).t()
            assert sel.shape[0] == n_sel, f"goal_selection must name {n_sel} goals per lane"
            rows = self.task_state[_L.TS_GOAL_LIST:_L.TS_GOAL_LIST + n_sel, :self._B]
            if m is None:
                rows.copy_(sel)
            else:
                rows.copy_(torch.where(m[:self._B].bool().unsqueeze(0), sel, rows))
        self.agents = copy.copy(self.agent_names)
        return self.get_observation(), self.infos

    def update(self, update_agents=False):
        """The task's own dynamics: the base class only has a clock (TaskEnvironment.py:353-359)."""
        self.t += self.dt
        self.history["t"].append(self.t)

    def step(self, actions=None, dt=None, drift_to_random_strength_ratio=1, *pos, **kws):
        """One closed-loop step (reference step, TaskEnvironment.py:361-453): move the agents with
        `drift_velocity = actions` ((B,2) device tensor / array, (2,) for all, a {name: array} dict,
        or None for random motion; NaNs count as 0), advance the task.  Returns device tensors
        `(observation (B,2), reward (B,) float64, terminal (B,) bool, truncated (B,) bool, infos)`.
        `agent_kwargs={...}` is forwarded to `Agent.update` (e.g. explicit `noise`)."""
        ag = self._agent
        if ag is None:
            raise AttributeError("Action is given, but there are no active agents. If there are no agents, try "
                                 "adding an agent with .add_agents().")
        agent_kwargs = kws.pop("agent_kwargs", {})
        if isinstance(actions, dict):
            actions = actions[self.agent_names[0]]
        if isinstance(drift_to_random_strength_ratio, dict):
            drift_to_random_strength_ratio = drift_to_random_strength_ratio[self.agent_names[0]]
        if actions is not None:
            if torch.is_tensor(actions):
                actions = torch.nan_to_num(actions, nan=0.0)
            else:
                actions = np.array(actions, dtype=np.float64)
                actions[np.isnan(actions)] = 0
        dt = dt if dt is not None else ag.dt
        ag.update(dt=dt, drift_velocity=actions, drift_to_random_strength_ratio=drift_to_random_strength_ratio,
                  **agent_kwargs)
        self.update(*pos, **kws)
        if self.goal_cache.goalorder == "custom":   # (after the agents have moved: where the reference's check raises)
            raise ValueError("Unknown mode: {}".format(self.goal_cache.goalorder))
        dev = ag.state_tensor.device
        env_s, walls = self.device_tables(dev)
        task = self._task_struct()
        st = ag.state_tensor
        if self._shared:
            rc = _L.lib.riab_task_world_step(env_s, task, _L.ptr(self.task_state), _L.ptr(self._world), _L.ptr(st[0]),
                                             _L.ptr(st[1]), self._B, float(self.t), _L.ptr(self._reward),
                                             _L.ptr(self._terminal), _L.ptr(self._met), _L.ptr(self._cand),
                                             _L.ptr(self._ticket), _L.ptr(self._diag), _L.current_stream())
            _L.check(rc, "riab_task_world_step")
        else:
            rc = _L.lib.riab_task_step(env_s, task, _L.ptr(self.task_state), _L.ptr(st[0]), _L.ptr(st[1]), self._B,
                                       float(self.t), _L.ptr(self._reward), _L.ptr(self._terminal), _L.ptr(self._diag),
                                       _L.current_stream())
            _L.check(rc, "riab_task_step")
        self._keep_step = (walls, task)
        return (self.get_observation(), self._reward, self._terminal.bool(), self._truncated, self.infos)

    def make_step_plan(self, neurons=None, capacity=1024, auto_reset=True, scripted_speed=None):
        """The whole closed-loop step as ONE native call (plan.py, riab_plan_*): `plan.step(1, drift_velocity=
        actions)` == `env.step(actions)` + `Neurons.update()` of every population (+ `env.reset(mask=terminal)`
        when `auto_reset`).  Read `env.get_reward()`, `env.terminal`, `env.get_observation()` afterwards.  With
        `lanes="agents"` and whole 256-agent segments the step is ONE kernel as well (csrc/riab_step1.hip, TASK & 8: the
        writer workgroups keep the world's books, the one that takes the last ticket walks the shared list, resets the world
        when its episode ended and posts the verdict; the store-bound populations ride along); other batches take three
        launches inside the one native call: motion + the world's step, its reset when the episode ended (decided on the
        device) + the next scripted action, the populations."""
        plan = self._agent.make_step_plan(neurons, capacity)
        return plan.attach_task(self, auto_reset=auto_reset, scripted_speed=scripted_speed)

    @property
    def terminal(self):
        """Terminal flag of every lane as of the last step (device bool (B,))."""
        return self._terminal.bool()

    def step1(self, action=None, *pos, **kws):
        """Single-lane shortcut returning python values (reference step1, TaskEnvironment.py:455-462)."""
        assert self._B == 1, "step1 is for a single agent"
        obs, rew, term, trunc, info = self.step(action, *pos, **kws)
        return [obs[0].cpu().numpy(), float(rew[0].item()), bool(term[0].item()), bool(trunc[0].item()),
                info[self.agent_names[0]]]

    # ---- readouts ---------------------------------------------------------------------------------
    def get_observation(self):
        """Observation of every lane: by default the positions, a (B,2) float64 device view."""
        if self._agent is None:
            return {}
        return self.observation_lambda[self.agent_names[0]](self._agent)

    def set_observation(self, agents=None, spaces=None, observation_lambdass=None):
        """Replace the observation function (takes the batched Agent, returns a tensor with a leading
        lane axis, e.g. `lambda ag: PCs.firingrate_tensor.t()`) and its space descriptor."""
        if isinstance(spaces, list):
            spaces = spaces[0]
        if isinstance(observation_lambdass, list):
            observation_lambdass = observation_lambdass[0]
        name = self.agent_names[0]
        self.observation_spaces[name] = spaces
        self.observation_lambda[name] = observation_lambdass

    def get_reward(self):
        """Reward total of every lane as of the last step (device float64 (B,))."""
        return self._reward[:self._B]

    @property
    def goals_left(self):
        if self._shared:   # (one list: the same count for every agent)
            return self._world[_L.TW_N_GOALS].long().expand(self._B)
        return self.task_state[_L.TS_N_GOALS, :self._B].long()

    @property
    def episode(self):
        """Episode counter per lane (device int tensor); the reference's scalar when B == 1 or the lanes share one world."""
        if self._shared:
            return int(self._world[_L.TW_EPISODE].item())
        e = self.task_state[_L.TS_EPISODE, :self._B].long()
        return int(e[0].item()) if self._B == 1 else e

    def _goal_vector(self, scale):
        st = self._agent.state_tensor
        out = torch.empty((2, self._B), dtype=torch.float64, device=st.device)
        env_s, walls = self.device_tables(st.device)
        task = self._task_struct()
        if self._shared:
            rc = _L.lib.riab_task_world_goal_vector(env_s, task, _L.ptr(self.task_state), _L.ptr(self._world), _L.ptr(st[0]),
                                                    _L.ptr(st[1]), self._B, float(scale), _L.ptr(out[0]), _L.ptr(out[1]),
                                                    _L.current_stream())
        else:
            rc = _L.lib.riab_task_goal_vector(env_s, task, _L.ptr(self.task_state), _L.ptr(st[0]), _L.ptr(st[1]), self._B,
                                              float(scale), _L.ptr(out[0]), _L.ptr(out[1]), _L.current_stream())
        _L.check(rc, "riab_task_goal_vector")
        self._keep_gv = (walls, task)
        return out.t()

    def _goal_lists(self):
        if self._shared:   # every agent's list is the shared one
            w = self._world.cpu().numpy()
            row = np.full(_L.TASK_MAX_GOALS, -1)
            n = int(w[_L.TW_N_GOALS])
            row[:n] = w[_L.TW_GOAL_LIST:_L.TW_GOAL_LIST + n].astype(int)
            return np.tile(row, (self._B, 1))
        ts = self.task_state[:, :self._B].cpu().numpy()
        n = ts[_L.TS_N_GOALS].astype(int)
        L = ts[_L.TS_GOAL_LIST:_L.TS_GOAL_LIST + _L.TASK_MAX_GOALS].T.astype(int)
        L[np.arange(_L.TASK_MAX_GOALS)[None, :] >= n[:, None]] = -1
        return L

    def _lanes_in_goal(self, goal):
        st = self._agent.state_tensor
        pos = st[0:2, :self._B].t()
        g = torch.as_tensor(np.asarray(goal.pos, dtype=np.float64), device=st.device)
        d = torch.linalg.norm(pos - g, dim=1)
        blocked = torch.zeros(self._B, dtype=torch.bool, device=st.device)
        for w in np.asarray(self.walls, dtype=np.float64)[4:]:
            blocked |= _segments_cross(pos, g, torch.as_tensor(w, device=st.device))
        return (d < goal.radius) & ~blocked

    @property
    def diagnostics(self):
        d = self._diag.cpu().numpy()
        return dict(reward_overflow=int(d[0]), late_completions=int(d[1]), episode_log_overflow=int(d[2]), resets=int(d[3]))

    @property
    def episodes(self):
        """Finished episodes of all lanes (synchronises): lists like the reference's `episodes` dict
        (TaskEnvironment.py:130-136) plus the global `lane` id each belongs to (-1: the one episode table of a world
        whose lanes are its agents, `lanes="agents"`)."""
        n = min(int(self._ep_count.item()), self._ep_cap) if self._agent is not None else 0
        log = self._ep_log[:n].cpu().numpy() if n else np.zeros((0, 5))
        order = np.lexsort((log[:, 0], log[:, 3])) if n else []
        log = log[order]
        return {"lane": log[:, 0].astype(int).tolist(), "episode": log[:, 1].astype(int).tolist(),
                "start": log[:, 2].tolist(), "end": log[:, 3].tolist(), "duration": log[:, 4].tolist()}

    def render(self, *a, **k):
        raise NotImplementedError("rendering is outside the accelerated path")

    def close(self):
        pass


def _segments_cross(p, g, wall):
    """Strict crossing of the segments p[i]->g with `wall` (utils.vector_intercepts logic)."""
    sa = g - p
    sb = wall[1] - wall[0]
    d0 = wall[0] - p
    den = sa[:, 0] * (-sb[1]) + sa[:, 1] * sb[0]
    la = (d0[:, 0] * (-sb[1]) + d0[:, 1] * sb[0]) / den
    lb = ((-d0[:, 0]) * (-sa[:, 1]) + (-d0[:, 1]) * sa[:, 0]) / (sb[0] * (-sa[:, 1]) + sb[1] * sa[:, 0])
    return (la > 0) & (la < 1) & (lb > 0) & (lb < 1)


class SpatialGoalEnvironment(TaskEnvironment):
    """A spatial goal-directed task (reference SpatialGoalEnvironment, TaskEnvironment.py:1376-1493)."""

    default_params = {}

    def __init__(self, *pos, possible_goals=None, possible_goal_positions="random_5", current_goal_state=None,
                 goalkws=dict(), **kws):
        super().__init__(*pos, **kws)
        self.goalkws = goalkws
        if possible_goals is None:
            self.goal_cache.reset_goals = self._init_poss_goal_positions(possible_goal_positions)
        else:
            self.goal_cache.reset_goals = possible_goals

    def _init_poss_goal_positions(self, possible_goal_position):
        """Pool of SpatialGoals from an array of positions or "random_<n>" (TaskEnvironment.py:1417-1456)."""
        if isinstance(possible_goal_position, str):
            if not possible_goal_position.startswith("random"):
                raise ValueError("possible_goal_pos string must start with 'random'")
            n = int(possible_goal_position.split("_")[1])
            ext = [self.extent[i:i + 2] for i in np.arange(0, len(self.extent), 2)]
            possible_goal_position = np.array([np.random.random(n) * (ext[i][1] - ext[i][0]) + ext[i][0]
                                               for i in range(len(ext))]).T
        possible_goal_position = np.array(possible_goal_position)
        return [SpatialGoal(self, pos=p, **self.goalkws) for p in possible_goal_position]

    def get_goal_positions(self):
        """(n_pool, 2) positions of the goals in the pool."""
        return np.array([g.pos for g in self.goal_cache.get_goals() if isinstance(g, SpatialGoal)])

    def reset(self, goal_locations=None, n_objectives=None, **kws):
        if goal_locations is not None:
            self.goal_cache.reset_n_goals = len(goal_locations)
        elif n_objectives is not None:
            self.goal_cache.reset_n_goals = n_objectives
        if goal_locations is not None:
            self.goal_cache.reset_goals = self._init_poss_goal_positions(goal_locations)
        return super().reset(**kws)


def get_goal_vector(Ag=None):
    """Vector from every lane's position to its goal: the head of the lane's list when goals are
    sequential, the nearest pending goal otherwise; zeros where a lane has no spatial goal pending
    (reference get_goal_vector, TaskEnvironment.py:1555-1584).  (B,2) float64 device tensor."""
    if isinstance(Ag, Agent):
        return Ag.Environment._goal_vector(0.0)
    if isinstance(Ag, list) and Ag and isinstance(Ag[0], Agent):   # (the reference's list / dict forms, :1575-1582)
        return {a.name: get_goal_vector(a) for a in Ag}
    if isinstance(Ag, dict):
        return {name: get_goal_vector(a) for name, a in Ag.items()}
    raise TypeError("Unknown input type")
