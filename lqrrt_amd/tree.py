"""
Tree -- the reference's node store (lqrrt/tree.py) as a view over the device-resident SoA
tree held by the HIP engine.

Same features as tree.py:1-27: `state` (N x n array), `pID`, `lqr` [(S, K)], `x_seq`, `u_seq`
(edge trajectories, parent -> node, parent excluded), `size`, `nstates`, `ncontrols`, plus
`climb` and `trajectory`.  Values are copied from HBM on access (cached until the tree
grows), so unlike the reference they do not alias the planner's storage.
"""
import numpy as np


class _EdgeSeq(object):
    """Lazy sequence: seq[ID] -> list of arrays along the edge into node ID (tree.py:21-25)."""

    def __init__(self, tree, which):
        self._tree, self._which = tree, which

    def __len__(self):
        return self._tree.size

    def __getitem__(self, ID):
        if isinstance(ID, slice):
            return [self[i] for i in range(*ID.indices(len(self)))]
        ID = int(ID)
        if ID < 0:
            ID += len(self)
        return self._tree._edge(ID)[self._which]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _LqrSeq(object):
    """Lazy sequence of (S, K) per node (tree.py:67,91)."""

    def __init__(self, tree):
        self._tree = tree

    def __len__(self):
        return self._tree.size

    def __getitem__(self, ID):
        ID = int(ID)
        if ID < 0:
            ID += len(self)
        base = self._tree._e.size
        if ID >= base:
            return self._tree._extra[ID - base][2]
        return (self._tree._S, self._tree._gains()[ID])


class Tree:
    """
    Device-backed tree.  The reference constructs Tree(seed_state, seed_lqr) inside
    update_plan (planner.py:172); here the planner binds it to its engine, whose
    lqrrt_tree_reset() kernel creates the seed node (pID -1, edge [[seed]], [[0]]).
    """

    def __init__(self, engine):
        self._e = engine
        self.nstates = engine.n
        self.ncontrols = engine.m
        self._S = engine.system.Smatrix()
        self._cache = {}
        self._cache_size = -1
        self._extra = []          # host-side nodes appended after planning (finish_on_goal, planner.py:299)
        self.x_seq = _EdgeSeq(self, 0)
        self.u_seq = _EdgeSeq(self, 1)
        self.lqr = _LqrSeq(self)

    @property
    def size(self):
        return self._e.size + len(self._extra)

    def _fresh(self):
        n = self._e.size
        if n != self._cache_size:
            self._cache = {}
            self._cache_size = n
        return self._cache

    @property
    def state(self):
        c = self._fresh()
        if "state" not in c:
            c["state"] = self._e.states()
        if self._extra:
            return np.vstack([c["state"]] + [np.asarray(x[1], dtype=np.float64) for x in self._extra])
        return c["state"]

    @property
    def pID(self):
        c = self._fresh()
        if "pID" not in c:
            c["pID"] = self._e.parents().tolist()
        if self._extra:
            return c["pID"] + [int(x[0]) for x in self._extra]
        return c["pID"]

    def _gains(self):
        c = self._fresh()
        if "K" not in c:
            c["K"] = self._e.gains()
        return c["K"]

    def _edge(self, ID):
        c = self._fresh()
        if ID >= self._e.size and ID - self._e.size < len(self._extra):
            x = self._extra[ID - self._e.size]
            return (x[3], x[4])
        key = ("edge", ID)
        if key not in c:
            if ID >= self.size or ID < 0:
                raise IndexError("node %d does not exist" % ID)
            x, u = self._e.edge(ID)
            c[key] = ([row for row in x], [row for row in u])
        return c[key]

    def add_node(self, pID, state, lqr, x_seq, u_seq):
        """
        tree.py:77-96.  During planning nodes are appended on the device by the waves; this host-side
        append exists for what the reference adds afterwards (the finish_on_goal node, planner.py:299).
        """
        if pID >= self.size or pID < 0:
            raise ValueError("The given parent ID, {}, doesn't exist.".format(pID))
        self._extra.append((pID, state, lqr, x_seq, u_seq))

    def climb(self, ID):
        """
        Returns a list of node IDs that connect the seed to the node with the given ID
        (first element 0, last element ID) -- tree.py:100-117.
        """
        if ID >= self.size or ID < 0:
            raise ValueError("The given ID, {}, doesn't exist.".format(ID))
        parents = self.pID
        IDs = []
        while ID != -1:
            IDs.append(int(ID))
            ID = parents[ID]
        return IDs[::-1]

    def trajectory(self, IDs):
        """Concatenated (x_seq_full, u_seq_full) over the listed nodes -- tree.py:121-132."""
        x_seq_full = []
        u_seq_full = []
        for ID in IDs:
            xs, us = self._edge(int(ID))
            x_seq_full.extend(xs)
            u_seq_full.extend(us)
        return (x_seq_full, u_seq_full)

    def visualize(self, dx, dy, node_seq=None):
        """Cross-section plot of the tree (tree.py:136-171); needs matplotlib."""
        from matplotlib import pyplot as plt
        fig = plt.figure()
        fig.suptitle('Tree')
        ax = fig.add_subplot(1, 1, 1)
        ax.set_xlabel('- State {} +'.format(dx))
        ax.set_ylabel('- State {} +'.format(dy))
        ax.grid(True)
        path = set(node_seq or [])
        parents = self.pID
        st = self.state
        for ID in range(1, self.size):
            xs = np.vstack((st[parents[ID]], np.array(self._edge(ID)[0])))
            ax.plot(xs[:, dx], xs[:, dy], color='r' if ID in path else '0.75', zorder=2 if ID in path else 1)
        ax.scatter(st[0, dx], st[0, dy], color='b', s=48)
        if node_seq:
            ax.scatter(st[node_seq[-1], dx], st[node_seq[-1], dy], color='r', s=48)
        plt.show()
