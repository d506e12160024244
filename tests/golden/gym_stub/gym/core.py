from . import error


class Env(object):
    """Old-gym (<0.9.6) dispatcher: public step/reset/render/seed/close forward
    to the underscore-prefixed implementations of the subclass."""
    metadata = {'render.modes': []}
    reward_range = (-float('inf'), float('inf'))
    action_space = None
    observation_space = None

    def step(self, action):
        return self._step(action)

    def reset(self):
        return self._reset()

    def render(self, mode='human', close=False):
        if not close:
            modes = self.metadata.get('render.modes', [])
            if len(modes) == 0:
                raise error.UnsupportedMode('{} does not support rendering'.format(self))
            if mode not in modes:
                raise error.UnsupportedMode('Unsupported rendering mode: {}'.format(mode))
        return self._render(mode=mode, close=close)

    def seed(self, seed=None):
        return self._seed(seed)

    def close(self):
        return self._close()

    def _close(self):
        pass

    def _seed(self, seed=None):
        return []
