"""Helpers shared by the SFMA tests: fixture access and trace comparison."""
import ast

import numpy as np

SFMA_WORLD_KEYS = ('next', 'reward', 'terminal', 'starts', 'coordinates', 'height', 'width',
                   'invalid_transitions')


def sfma_cases(z):
    return sorted({k.split('/')[0] for k in z.files if not k.startswith(('world/', 'metric/'))})


def sfma_case(z, name):
    """(record accessor, world tables, metric matrix, options) of one golden run."""
    g = lambda k: z['%s/%s' % (name, k)]          # noqa: E731
    opts = ast.literal_eval(str(g('opts')))
    wname = opts['world']
    world = {k: z['world/%s/%s' % (wname, k)] for k in SFMA_WORLD_KEYS}
    D = z['metric/%s/%s' % (wname, opts['metric'])]
    if opts.get('mask'):
        opts['mask'] = g('action_mask')
    return g, world, D, opts
