"""Host-side data plumbing with the interface of the reference's ``colvarsfinder.utils`` (own code).

Not on the accelerated path: these helpers only exist so that the reference's
2D example (``examples/2d/2d.ipynb:280,395,485``) runs unchanged against this package.

* ``WeightedTrajectory``        <- utils.py:62-169   trajectory + weights holder (MDAnalysis universe, duck-typed,
                                                     or a text file ``time x_1 .. x_d`` per line)
* ``integrate_sde_overdamped``  <- utils.py:257-352  Euler-Maruyama sampler, writes ``traj.txt`` / ``output.csv``
* ``calc_weights``              <- utils.py:354-417  re-weighting factors exp(-(beta_sys-beta_sim)(V - mean V)), mean 1
* ``integrate_md_langevin``     <- utils.py:172-255  OpenMM driver: not provided (wraps an external MD engine)
* ``RowReader`` / ``MappedTrajectory``  (no counterpart in the reference, which holds the whole array in memory, utils.py:106):
                                                     a trajectory served from a memory-mapped ``.npy`` file row by row, so that a
                                                     rank of a data-parallel job reads only the frames of its own batch slices
"""

import math
import os

import numpy as np
import pandas as pd


class WeightedTrajectory:
    """Trajectory array ``trajectory`` ([n, N, 3] from a universe, [n, d] from text), ``weights`` (mean one),
    ``dt`` and ``n_frames``; states whose normalised weight is outside (min_w, max_w) are dropped."""

    def __init__(self, universe=None, input_ag=None, traj_filename=None, weight_filename=None, min_w=0.0,
                 max_w=float("inf"), verbose=True):
        if universe is not None:
            atoms = universe.atoms.ix if input_ag is None else input_ag.ix
            self.trajectory = universe.trajectory.timeseries(order='fac')[:, atoms, :]
            self.n_frames = universe.trajectory.n_frames
            self.dt = universe.trajectory.dt * 1e-3   # ps -> ns
            if verbose:
                print(f'\nTrajectory Info:\n  no. of frames in trajectory data: {self.n_frames}\n'
                      f'  stepsize: {universe.trajectory.dt:.1f}ps\n  shape of trajectory data array: {self.trajectory.shape}\n')
        else:
            if traj_filename is None or not os.path.exists(traj_filename):
                raise FileNotFoundError('trajectory file not found')
            table = np.loadtxt(traj_filename)
            self.n_frames = table.shape[0]
            self.trajectory = table[:, 1:]
            self.dt = table[1, 0] - table[0, 0]

        if not weight_filename:
            self.weights = np.ones(self.n_frames)
            return
        w = pd.read_csv(weight_filename, usecols=[0], header=None)[0]
        w = w / w.mean()
        if verbose:
            print('\nloading weights from file: ', weight_filename)
            print('\nWeights:\n', w.describe(percentiles=[0.2, 0.4, 0.6, 0.8]))
        if self.n_frames != len(w.index):
            raise ValueError('length in weight file does match the trajectory data!\n')
        keep = (w > min_w) & (w < max_w)
        self.trajectory = self.trajectory[keep.to_numpy(), ...]
        kept = w[keep]
        kept = kept / kept.mean()
        if verbose:
            print(f'\nAfter selecting states whose weights are in [{min_w:.3e}, {max_w:.3e}] and renormalization:\n'
                  f'\nShape of trajectory: {self.trajectory.shape}')
            print('\nWeights:\n', kept.describe(percentiles=[0.2, 0.4, 0.6, 0.8]))
        self.weights = kept.to_numpy()


def integrate_md_langevin(*args, **kwargs):
    raise NotImplementedError("integrate_md_langevin drives OpenMM; it is outside the scope of the MI355X hot-path build")


def integrate_sde_overdamped(pot_obj, n_steps, sampling_output_path, X0=None, pre_steps=0, step_size=0.01,
                             traj_txt_filename='traj.txt', csv_filename='output.csv', report_interval=100,
                             report_interval_stdout=100):
    """Euler-Maruyama for dX = -grad V dt + sqrt(2/beta) dW.  ``pot_obj`` has ``dim``, ``beta``, ``V``, ``gradV``.
    Every ``report_interval`` steps a line ``time x_1 .. x_d`` (``%.3f`` / ``%.6f``) goes to the text file and
    ``(time, energy)`` to the CSV (header ``Time,Energy``)."""
    dim, beta = pot_obj.dim, pot_obj.beta
    print(f'Directory to save trajectory ouptuts: {sampling_output_path}')
    print(f'sampling beta={beta:.3f}, dt={step_size:.3f}\n')
    x = np.random.randn(dim) if X0 is None else X0
    noise = np.sqrt(2 * step_size / beta)

    def advance(state):
        xi = np.random.randn(dim)
        return state - pot_obj.gradV(state) * step_size + noise * xi

    print(f'First, burning, total number of steps = {pre_steps}')
    for _ in range(pre_steps):
        x = advance(x)
    print(f'Next, run {n_steps} steps')
    rows = []
    with open(os.path.join(sampling_output_path, traj_txt_filename), 'w+') as out:
        for i in range(n_steps):
            x = advance(x)
            if i % report_interval == 0:
                out.write(f"{i * step_size:.3f} " + ' '.join(f'{v:.6f}' for v in x) + '\n')
                rows.append([i * step_size, pot_obj.V(x)])
            if i % report_interval_stdout == 0:
                print(f'step={i}, time={i * step_size:.3f}, energy={pot_obj.V(x):.3f}', flush=True)
    pd.DataFrame(rows, columns=['Time', 'Energy']).to_csv(os.path.join(sampling_output_path, csv_filename), index=False)


def calc_weights(csv_filename, sampling_beta, sys_beta, traj_weight_filename='weights.txt', energy_col_idx=1):
    """Weights ``exp(-(sys_beta - sampling_beta) (V_i - mean V))`` normalised to mean one, one per line."""
    print('\n=============Calculate Weights============')
    print(f'Reading potential from: {csv_filename}')
    table = pd.read_csv(csv_filename)
    table.rename(columns={table.columns[0]: 'Time'}, inplace=True)
    col = table.columns[energy_col_idx]
    print('\nUse {:d}th column to reweight, name: {}'.format(energy_col_idx, col))
    energy = table[col]
    mean_energy = energy.mean()
    print(f'\nsampling beta={sampling_beta}, system beta={sys_beta}')
    raw = [math.exp(-(sys_beta - sampling_beta) * (e - mean_energy)) for e in energy]
    weights = pd.DataFrame(raw / np.mean(raw), columns=['weight'])
    print('\nSummary of weights:\n', weights.describe())
    weights.to_csv(traj_weight_filename, header=False, index=False)
    print(f'weights saved to: {traj_weight_filename}')


class RowReader:
    """Frames ``[n, ...]`` behind an array-like that is read ROW-WISE (an ``np.memmap`` of a ``.npy`` file, an HDF5 dataset, ..):
    the training tasks call ``take_rows(rows)`` for exactly the frames they keep resident - in a data-parallel job a rank's
    slices of the static batches, 1/world of the trajectory - instead of holding the whole array in
    host memory on every rank (the reference's ``WeightedTrajectory`` does, utils.py:106; at BASELINE config 5 that is 120 GB
    per rank).  ``rows_read`` / ``bytes_read`` count what was asked for (tools/check_dp2.py asserts the 1/world share)."""

    def __init__(self, source):
        self._src = source
        self.shape = tuple(source.shape)
        self.dtype = source.dtype
        self.rows_read = 0
        self.bytes_read = 0

    def __len__(self):
        return self.shape[0]

    def take_rows(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        order = np.argsort(rows, kind="stable")          # ascending file offsets: one forward sweep over the pages
        out = np.empty((len(rows),) + self.shape[1:], dtype=self.dtype)
        step = max(1, (64 << 20) // max(1, int(np.prod(self.shape[1:])) * self.dtype.itemsize))
        for s0 in range(0, len(rows), step):              # bounded temporaries whatever the shard size
            sel = order[s0:s0 + step]
            out[sel] = self._src[rows[sel]]
        self.rows_read += len(rows)
        self.bytes_read += out.nbytes
        return out

    def __getitem__(self, key):   # (row 0 for the shapes the constructors look at; anything else reads what it names)
        return np.asarray(self._src[key])


class MappedTrajectory:
    """``traj_obj`` over a memory-mapped ``.npy`` trajectory: ``trajectory`` is a :class:`RowReader`, ``weights`` (mean one;
    default all ones) and ``dt`` as in ``WeightedTrajectory``.  A single-process task reads every frame once (the upload);
    a rank of a data-parallel job reads its own rows only."""

    def __init__(self, traj_npy, weights=None, dt=1.0):
        self.trajectory = RowReader(np.load(traj_npy, mmap_mode="r"))
        self.n_frames = self.trajectory.shape[0]
        self.weights = np.ones(self.n_frames) if weights is None else np.asarray(weights, dtype=np.float64)
        assert len(self.weights) == self.n_frames, 'length of weights does not match the trajectory data'
        self.dt = dt
