"""Headless stand-in for the reference's pyglet renderer (rendering.py:50-336, out of scope
as a GUI): records what that window would have drawn -- poses, speeds, lap info, optionally
the scans -- for any subset of a batch, on the device, and dumps them to .npz for offline
replay (the reference's own recorder is f1tenth_gym/examples/lidar.py:212-254)."""
import numpy as np
import torch


class TrajectoryRecorder(object):
    def __init__(self, vec_env, max_steps, envs=None, with_scans=False):
        self.env = vec_env
        dev = vec_env.device
        self.idx = torch.arange(vec_env.num_envs, device=dev) if envs is None else torch.as_tensor(envs, device=dev)
        n, A = self.idx.numel(), vec_env.num_agents
        self.max_steps, self.t = int(max_steps), 0
        self.state = torch.zeros((self.max_steps, n, A, 7), dtype=torch.float64, device=dev)
        self.collisions = torch.zeros((self.max_steps, n, A), dtype=torch.uint8, device=dev)
        self.lap_counts = torch.zeros((self.max_steps, n, A), dtype=torch.int32, device=dev)
        self.lap_times = torch.zeros((self.max_steps, n, A), dtype=torch.float64, device=dev)
        self.done = torch.zeros((self.max_steps, n), dtype=torch.bool, device=dev)
        self.scans = (torch.zeros((self.max_steps, n, A, vec_env.eng.num_beams), dtype=torch.float32, device=dev)
                      if with_scans else None)

    def record(self):
        """Call after reset()/step(): copies the current observation of the selected envs."""
        if self.t >= self.max_steps:
            raise IndexError('TrajectoryRecorder is full (%d steps)' % self.max_steps)
        t = self.env.eng.t
        k = self.t
        self.state[k] = t['state'][self.idx]
        self.collisions[k] = t['collisions'][self.idx]
        self.lap_counts[k] = t['lap_counts'][self.idx]
        self.lap_times[k] = t['lap_times'][self.idx]
        self.done[k] = t['done'][self.idx]
        if self.scans is not None:
            self.scans[k] = t['scans'][self.idx]
        self.t += 1

    def arrays(self):
        out = {'env_index': self.idx.cpu().numpy(), 'state': self.state[:self.t].cpu().numpy(),
               'collisions': self.collisions[:self.t].cpu().numpy(), 'lap_counts': self.lap_counts[:self.t].cpu().numpy(),
               'lap_times': self.lap_times[:self.t].cpu().numpy(), 'done': self.done[:self.t].cpu().numpy(),
               'timestep': np.float64(self.env.timestep)}
        if self.scans is not None:
            out['scans'] = self.scans[:self.t].cpu().numpy()
        return out

    def save(self, path):
        np.savez_compressed(path, **self.arrays())
        return path


class LidarDatasetWriter(object):
    """The reference's lidar dataset recorder (f1tenth_gym/examples/lidar.py:158-254) for a batch of envs: every
    step's ego scan becomes a 256x256 point-occupancy grid (`occupancy_kernel`, :212-244) that is appended to a
    device buffer; `save()` writes what the reference writes -- `np.savez_compressed(file, data=uint8 [N,256,256])`
    (:250-254), values 0 / 1 -- so the reference's consumers read the files unchanged.

    `add(scans)` takes the ego scans of a step, [B, num_beams] on the device (fp32 as `f110_step` writes them, or
    fp64), and an optional mask [B] of the envs still recording: the reference stops an episode at `done` (:198).
    `record_episodes` is the reference's own loop (random spawn around the origin, 10 random-action steps per
    episode, :186-206) with one episode per env."""

    def __init__(self, device, capacity, grid_size=256, max_range=30.0, lo=-10.0, hi=10.0):
        self.device = torch.device(device)
        self.grid, self.max_range, self.lo, self.hi = int(grid_size), float(max_range), float(lo), float(hi)
        self.data = torch.zeros((int(capacity), self.grid, self.grid), dtype=torch.uint8, device=self.device)
        self.n = 0

    def add(self, scans, mask=None):
        from .lidar import scan_occupancy
        s = scans if mask is None else scans[mask.to(device=scans.device, dtype=torch.bool)]
        if s.shape[0] == 0:
            return 0
        if self.n + s.shape[0] > self.data.shape[0]:
            raise IndexError('LidarDatasetWriter is full (%d frames)' % self.data.shape[0])
        grids = scan_occupancy(s.contiguous(), self.max_range, self.lo, self.hi, self.grid)
        self.data[self.n:self.n + s.shape[0]] = grids
        self.n += s.shape[0]
        return s.shape[0]

    def array(self):
        return self.data[:self.n].cpu().numpy()

    def save(self, path):
        """np.savez_compressed(path, data=...) exactly as lidar.py:252; returns the number of frames written."""
        np.savez_compressed(path, data=self.array())
        return self.n

    @staticmethod
    def record_episodes(vec_env, steps=10, seed=None, writer=None):
        """One episode per env of `vec_env` (1 agent, autoreset off): spawn x, y ~ U(-2, 2), theta ~ U(-pi, pi)
        (:186-189), then `steps` steps of steer ~ U(-0.5, 0.5), speed ~ U(0, 3) (:203-206); a frame per env and step
        until the env reports done (:198-199).  Frames are appended step-major ([step][env])."""
        B = vec_env.num_envs
        g = torch.Generator(device=vec_env.device)
        if seed is not None:
            g.manual_seed(int(seed))
        u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, generator=g, dtype=torch.float64, device=vec_env.device)  # noqa: E731
        poses = torch.stack([u(-2.0, 2.0, (B, 1)), u(-2.0, 2.0, (B, 1)), u(-np.pi, np.pi, (B, 1))], dim=-1)
        if writer is None:
            writer = LidarDatasetWriter(vec_env.device, B * steps)
        obs, _, done, _ = vec_env.reset(poses)
        alive = ~done.clone()
        for _ in range(steps):
            if not bool(alive.any()):
                break
            act = torch.stack([u(-0.5, 0.5, (B, 1)), u(0.0, 3.0, (B, 1))], dim=-1)
            obs, _, done, _ = vec_env.step(act)
            writer.add(obs['scans'][:, vec_env.ego_idx], alive)   # the step's scan is recorded, then `done` ends the episode
            alive &= ~done
        return writer
