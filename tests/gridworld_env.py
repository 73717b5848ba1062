"""A GridWorld-shaped environment in the reference's Python env protocol (python_interface/pyenv.rs): the dynamics of
examples/grid_world/src/lib.rs:84-164 -- agent / goal / trap on a width x height grid, actions up / down / left / right,
reward 1.0 at the goal, -0.5 at the trap or out of steps, else -0.5 / steps_left; obs id of cell i = i * (w*h) + {0 empty,
1 agent, 2 goal, 3 trap}.  Test fixture (user code from the collector's point of view: the SAME class runs under the HIP
collector and under the oracle).  Its placement draws are seeded per (seed, episode) so that a collect is reproducible."""
import random


class GridWorld:
    def __init__(self, width, height, max_steps):
        self.width, self.height, self.max_steps = width, height, max_steps
        self.steps_left = max_steps
        self.agent = self.goal = self.trap = (0, 0)
        self.rng = random.Random(0)
        self.max_records = max_steps + 1

    def copy(self):
        c = GridWorld(self.width, self.height, self.max_steps)
        c.steps_left, c.agent, c.goal, c.trap = self.steps_left, self.agent, self.goal, self.trap
        return c

    def seed_episode(self, seed, episode):
        self.rng = random.Random(seed * 1000003 + episode)

    def num_actions(self):
        return 4

    def obs_shape(self):
        return [self.width * self.height, self.width * self.height]

    def _random_pos(self):
        i = self.rng.randrange(self.width * self.height)
        return (i % self.width, i // self.width)

    def reset(self, difficulty):                                   # lib.rs:118-129
        difficulty = min(self.width + self.height, difficulty)
        self.agent = self._random_pos()
        near = [(x, y) for x in range(self.width) for y in range(self.height)
                if abs(x - self.agent[0]) + abs(y - self.agent[1]) <= difficulty]
        while True:
            g = self.rng.choice(near)
            if g != self.agent:
                break
        self.goal = g
        while True:
            t = self._random_pos()
            if t != self.agent and t != self.goal:
                break
        self.trap = t
        self.steps_left = self.max_steps

    def next(self, action):                                        # lib.rs:131-140
        x, y = self.agent
        if action == 0 and y > 0:
            y -= 1
        elif action == 1 and y + 1 < self.height:
            y += 1
        elif action == 2 and x > 0:
            x -= 1
        elif action == 3 and x + 1 < self.width:
            x += 1
        self.agent = (x, y)
        self.steps_left = max(0, self.steps_left - 1)

    def masks(self):                                               # lib.rs:142-149
        x, y = self.agent
        return [y > 0, y + 1 < self.height, x > 0, x + 1 < self.width]

    def is_final(self):
        return self.steps_left == 0 or self.agent == self.goal or self.agent == self.trap

    def value(self):                                               # lib.rs:155-159
        if self.agent == self.goal:
            return 1.0
        if self.agent == self.trap or self.steps_left == 0:
            return -0.5
        return -0.5 / float(self.steps_left)

    def success(self):
        return self.agent == self.goal

    def observe(self):                                             # lib.rs:165-167
        n = self.width * self.height
        board = [0] * n
        idx = lambda p: p[1] * self.width + p[0]
        board[idx(self.goal)] = 2
        board[idx(self.trap)] = 3
        board[idx(self.agent)] = 1
        return [i * n + v for i, v in enumerate(board)]

    def set_state(self, board):
        for i, v in enumerate(board):
            p = (i % self.width, i // self.width)
            if v == 1:
                self.agent = p
            elif v == 2:
                self.goal = p
            elif v == 3:
                self.trap = p
        self.steps_left = self.max_steps


class TrackingGridWorld(GridWorld):
    """A GridWorld that keeps its own record of the way it was solved -- `Env::track_solution` / `Env::solution`
    (rust/src/rl/env.rs:61-66): `single_solve` then returns THIS instead of the actions it played (rust/src/rl/solve.rs:28,
    57-64).  An entry is 1000 x (cell entered) + action: not an action index, and above 255 on purpose (the 32-bit path).
    It also names its symmetry as twists (mirror left <-> right): Env::twists (env.rs:58-59), which the host feeds the Policy."""

    def __init__(self, width, height, max_steps):
        super().__init__(width, height, max_steps)
        self.path = []

    def copy(self):
        c = TrackingGridWorld(self.width, self.height, self.max_steps)
        c.steps_left, c.agent, c.goal, c.trap = self.steps_left, self.agent, self.goal, self.trap
        c.path = list(self.path)
        return c

    def reset(self, difficulty):
        super().reset(difficulty)
        self.path = []

    def next(self, action):
        super().next(action)
        self.path.append(1000 * (self.agent[1] * self.width + self.agent[0]) + int(action))

    def track_solution(self):
        return True

    def solution(self):
        return list(self.path)

    def twists(self):
        n, w = self.width * self.height, self.width
        mirror = lambda i: (i // w) * w + (w - 1 - i % w)
        ident = list(range(n * n))
        mir = [mirror(o // n) * n + (o % n) for o in range(n * n)]
        return [ident, mir], [[0, 1, 2, 3], [0, 1, 3, 2]]
