"""Grid search (optimizer/grid_search.py:18-312 of the reference).

Same constructor, the same three enumeration orders of ``parameter_combinations``, the same
``fit`` / ``recompute_fit`` contract and on-disk files (``fit.pkl``, ``sim_<v1>_<v2>....pkl``), so
result directories are interchangeable.  The reference evaluates ``simulation(task, parameters)``
once per combination and run, sequentially or through a process pool (:219-247).  With 10^4-10^5
agent-environment instances resident on one GPU the natural unit is different: ``fit_vectorised``
hands ALL pending combinations to one call of a batched simulation, which lays them out on the
instance axis (``spread_over_instances``) and runs them in a single kernel launch with
per-instance hyper-parameters (``cobel_param_set_t``).
"""
from __future__ import annotations

import copy
import pickle
from itertools import product
from os import listdir
from os.path import isfile, join

import numpy as np

from .optimizer import Optimizer


def _file_name(combination: tuple) -> str:
    return 'sim' + ('_%s' * len(combination)) % combination + '.pkl'


def spread_over_instances(combinations: list[dict], nb_runs: int) -> tuple[dict, np.ndarray]:
    """Per-instance hyper-parameter arrays for ``len(combinations) * nb_runs`` instances
    (combination-major, run-minor) and, for each instance, the index of its combination.  The
    arrays go straight into an agent's ``learning_rate`` / ``gamma`` or a policy's ``epsilon``."""
    which = np.repeat(np.arange(len(combinations)), nb_runs)
    keys = list(combinations[0]) if combinations else []
    return {k: np.array([combinations[c][k] for c in which]) for k in keys}, which


class GridSearchOptimizer(Optimizer):
    def __init__(self, file_path: str, parameters: dict, nb_runs: int = 1, order: str = 'nested',
                 rng=None) -> None:
        self.parameters = copy.deepcopy(parameters)
        for name, values in self.parameters.items():
            if type(values) is np.ndarray:
                self.parameters[name] = np.sort(values)
        self.rng = np.random.default_rng() if rng is None else rng
        self.prepare_parameter_combinations(self.parameters, order)
        self.file_path = file_path
        self.nb_runs = nb_runs
        self.order = order
        self.present_files = [f for f in listdir(file_path) if isfile(join(file_path, f))]

    # -- enumeration (grid_search.py:112-171) ---------------------------------------------------
    def prepare_parameter_combinations(self, parameters: dict, order: str = 'nested') -> None:
        assert order in ['nested', 'shuffled', 'systematic'], 'Invalid order!'
        names = list(parameters)
        grids: list = []
        if order == 'systematic':
            # coarse-to-fine: level i visits every stride_i-th value of each parameter, the strides
            # halving from len/2 down to 1; combinations seen at a coarser level keep their place
            ladders = []
            for name in names:
                count = len(parameters[name])
                ladder = [max(int(count / (2 ** (i + 1))), 1)
                          for i in range(int(np.ceil(np.sqrt(count))))]
                if 1 not in ladder:
                    ladder.append(1)
                ladders.append(ladder)
            depth = max(len(ladder) for ladder in ladders)
            ladders = [ladder + [1] * (depth - len(ladder)) for ladder in ladders]
            for level in range(depth):
                axes = []
                for name, ladder in zip(names, ladders):
                    values, stride = list(parameters[name]), ladder[level]
                    axes.append([values[v * stride] for v in range(len(values) // stride)])
                grids.append(product(*axes))
        else:
            if order == 'shuffled':
                for name in names:
                    self.rng.shuffle(parameters[name])
            grids.append(product(*parameters.values()))
        self.parameter_combinations: dict = {}
        for grid in grids:
            for combination in grid:
                key = tuple(combination)
                if key not in self.parameter_combinations:
                    self.parameter_combinations[key] = dict(zip(names, combination))

    # -- fitting (grid_search.py:173-262) ---------------------------------------------------------
    def _load_fit(self) -> dict:
        if 'fit.pkl' in self.present_files:
            with open(self.file_path + 'fit.pkl', 'rb') as fh:
                return pickle.load(fh)
        return {}

    def _finish(self, fit: dict, key: tuple, simulation_data: dict, data: dict, loss,
                store_simulation_data: bool) -> None:
        if store_simulation_data:
            with open(self.file_path + _file_name(key), 'wb') as fh:
                pickle.dump(simulation_data, fh)
        fit[key] = loss(simulation_data, data)
        with open(self.file_path + 'fit.pkl', 'wb') as fh:
            pickle.dump(fit, fh)

    def _pending(self, fit: dict, overwrite: bool):
        """(key, stored simulation data or None) for every combination still to be fitted."""
        for key in self.parameter_combinations:
            if key in fit and not overwrite:
                continue
            stored = None
            if _file_name(key) in self.present_files and not overwrite:
                with open(self.file_path + _file_name(key), 'rb') as fh:
                    stored = pickle.load(fh)
            yield key, stored

    def fit(self, simulation, tasks: dict, data: dict, loss, overwrite: bool = False,
            store_simulation_data: bool = False, pool=None) -> dict:
        assert tasks.keys() == data.keys(), 'Task mismatch!'
        fit = self._load_fit()
        for key, stored in self._pending(fit, overwrite):
            if stored is not None:
                # as in the reference, data found on disk is loaded but no fit is derived from it
                # here (recompute_fit does that)
                continue
            params = self.parameter_combinations[key]
            simulation_data = {}
            for task in tasks:
                if pool is None:
                    simulation_data[task] = [simulation(tasks[task], params)
                                             for _ in range(self.nb_runs)]
                else:
                    jobs = [pool.apply_async(simulation, (tasks[task], params))
                            for _ in range(self.nb_runs)]
                    simulation_data[task] = [job.get() for job in jobs]
            self._finish(fit, key, simulation_data, data, loss, store_simulation_data)
        return fit

    def fit_vectorised(self, simulation_batch, tasks: dict, data: dict, loss,
                       overwrite: bool = False, store_simulation_data: bool = False) -> dict:
        """Like ``fit``, but all pending combinations of a task are simulated by ONE call
        ``simulation_batch(task, combinations, nb_runs)`` that returns, per combination, the list
        of its ``nb_runs`` results (what ``fit`` would have collected run by run)."""
        assert tasks.keys() == data.keys(), 'Task mismatch!'
        fit = self._load_fit()
        keys = [key for key, stored in self._pending(fit, overwrite) if stored is None]
        if not keys:
            return fit
        combos = [self.parameter_combinations[key] for key in keys]
        per_task = {}
        for task in tasks:
            results = simulation_batch(tasks[task], combos, self.nb_runs)
            assert len(results) == len(keys), 'one result list per parameter combination'
            per_task[task] = results
        for k, key in enumerate(keys):
            simulation_data = {task: list(per_task[task][k]) for task in tasks}
            self._finish(fit, key, simulation_data, data, loss, store_simulation_data)
        return fit

    def recompute_fit(self, data: dict, loss, overwrite: bool = False) -> dict:
        fit = self._load_fit()
        if not fit:
            if 'fit.pkl' not in self.present_files:
                print('No fit to recompute!')
            return fit
        for key in fit:
            if _file_name(key) in self.present_files:
                with open(self.file_path + _file_name(key), 'rb') as fh:
                    fit[key] = loss(pickle.load(fh), data)
            else:
                print('No simulation data found for parameter combination: ' + str(key))
            if overwrite:
                with open(self.file_path + 'fit.pkl', 'wb') as fh:
                    pickle.dump(fit, fh)
        return fit
