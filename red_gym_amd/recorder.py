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
