"""
Tree -- the reference's node store (lqrrt/tree.py) for a tree that lives on the GPU while it grows.

Features, as in tree.py:1-27: `state` (N x n array), `pID`, `lqr` [(S, K)], `x_seq` / `u_seq` (edge trajectories
parent -> node, parent excluded), `size`, `nstates`, `ncontrols`; methods `add_node`, `climb`, `trajectory`.

Two residences:
  * host    Tree(seed_state, seed_lqr) builds an ordinary host-side tree, exactly the reference's constructor
            (tree.py:50).  add_node appends to it.
  * device  During Planner.update_plan the planner binds the tree to its engine: the nodes are then the engine's
            SoA buffers in HBM and every feature is copied out on access (cached until the tree grows).  When the
            engine is about to be reused for the next plan the planner detaches the tree first, i.e. snapshots
            all features to the host in five bulk copies (kept as arrays: per-node rows are made on access, as for a
            bound tree -- a 35k-node tree detaches in milliseconds) -- so a Tree object kept from an earlier plan
            (the ROS node does that: lqrrt_node.py:477 keeps planner.tree and reads it at :889, :1149 while the
            next plan grows) stays what it was, and never touches the engine again.

Unlike the reference ("PASSES BY REFERENCE", tree.py:27) values read from a device-resident tree are copies.
"""
import numpy as np


class _Rows(object):
    """Read-only sequence over the nodes of a tree; item i is produced by `get(i)`."""

    def __init__(self, tree, get):
        self._tree, self._get = tree, get

    def __len__(self):
        return self._tree.size

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(k) for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("node %d does not exist" % i)
        return self._get(i)

    def __iter__(self):
        return (self._get(k) for k in range(len(self)))


class _Snapshot(object):
    """The nodes of a detached tree: the engine's five bulk copies behind the getters a bound Tree reads its engine through."""

    epoch = 0

    def __init__(self, e):
        self.size = e.size
        self._state, self._pid, self._K = e.states(), e.parents(), e.gains()
        # The edges leave HBM through page-locked staging (PCIe speed) and are kept COMPACTED in ordinary memory: rows beyond an edge's
        # length are dropped (a tree's edges fill ~40 % of the fixed-stride pools) and the staging buffers go back to torch's caching host
        # allocator, which hands the same blocks to the next detach -- a kept Tree pins nothing.
        xe, ue, ln = e.edges(pinned=True)
        live = np.arange(xe.shape[1])[None, :] < ln[:, None]
        self._xe, self._ue = xe[live], ue[live]
        self._off = np.concatenate(([0], np.cumsum(ln, dtype=np.int64)))
        del xe, ue

    def states(self):
        return self._state

    def parents(self):
        return self._pid

    def gains(self):
        return self._K

    def edge(self, i):
        a, b = int(self._off[i]), int(self._off[i + 1])
        return self._xe[a:b], self._ue[a:b]


def held_elsewhere(owner, attr):
    """Does anybody besides `owner.<attr>` still hold that tree or one of its feature sequences?  CPython reference counts, calibrated
    against a freshly built tree counted by the very same expression from the very same kind of name (a local of this function), so
    that nothing is assumed about the tree's references to itself or about the interpreter's call overhead.  If nobody does, the
    contents of a tree that is about to lose its engine can never be read again and need not be copied out of HBM.  (An interpreter
    without sys.getrefcount: "yes".)"""
    import sys
    if not hasattr(sys, "getrefcount"):
        return True

    def counts(t):
        return [sys.getrefcount(t)] + [sys.getrefcount(getattr(t, name)) for name in ("x_seq", "u_seq", "lqr")]
    tree = getattr(owner, attr)
    probe = Tree(np.zeros(tree.nstates), None)
    base, mine = counts(probe), counts(tree)      # both held by one local of this function; the tree also by the owner's attribute
    if mine[0] - base[0] > 1:
        return True
    return any(m > b for m, b in zip(mine[1:], base[1:]))


class Tree:
    """
    Tree(seed_state, seed_lqr): seed_state is the state of the root, seed_lqr the (S, K) tuple of local LQR
    cost-to-go and gain matrices there (tree.py:41-47).
    """

    def __init__(self, seed_state, seed_lqr):
        seed = np.array(seed_state, dtype=np.float64)
        self.nstates = len(seed)
        K0 = seed_lqr[1] if seed_lqr is not None else None
        self.ncontrols = int(np.shape(K0)[0]) if K0 is not None and np.ndim(K0) == 2 else 1     # tree.py:57-62
        # host residence: per-node python lists (the tree is small whenever it lives here while growing)
        self._h_state = [seed]
        self._h_pID = [-1]
        self._h_lqr = [seed_lqr]
        self._h_x = [[seed]]                                    # tree.py:69: the root's edge is the seed itself
        self._h_u = [[np.zeros(self.ncontrols)]]                # tree.py:70
        # device residence
        self._e = None
        self._snap = None                                       # nodes [0, snap.size) of a detached tree (host arrays)
        self._generation = None
        self._S = None
        self._cache = {}
        self._cache_size = -1
        self.x_seq = _Rows(self, lambda i: self._edge(i)[0])
        self.u_seq = _Rows(self, lambda i: self._edge(i)[1])
        self.lqr = _Rows(self, self._lqr_of)

    # -- residence ---------------------------------------------------------------------------------
    def _bind(self, engine, S):
        """The engine's tree (just reset to this tree's seed) becomes the storage of nodes [0, engine.size)."""
        self._e = engine
        self._snap = None
        self._generation = engine.generation
        self._S = S
        self._cache, self._cache_size = {}, -1
        self._h_state, self._h_pID, self._h_lqr, self._h_x, self._h_u = [], [], [], [], []     # host part: later nodes only

    @property
    def on_device(self):
        return self._e is not None

    def _dev(self):
        """The engine, after checking that it still holds THIS tree."""
        if self._e.generation != self._generation:
            raise RuntimeError("this Tree's engine was reset for another plan before the tree was detached")
        return self._e

    def _src(self):
        """Where nodes [0, _ndev()) live: the engine (after checking that it still holds THIS tree), a detached tree's snapshot, or None."""
        return self._dev() if self._e is not None else self._snap

    def _detach(self):
        """Snapshots every feature to the host (five bulk copies) and lets go of the engine."""
        if self._e is None:
            return
        self._snap = _Snapshot(self._dev())
        self._e, self._generation, self._cache, self._cache_size = None, None, {}, -1

    # -- features -----------------------------------------------------------------------------------
    def _ndev(self):
        src = self._src()
        return src.size if src is not None else 0

    @property
    def size(self):
        return self._ndev() + len(self._h_pID)

    def _fresh(self):
        # keyed on (size, epoch): a truncate or rewind followed by regrowth to the same size must not serve old nodes
        key = (self._ndev(), getattr(self._src(), "epoch", 0))
        if key != self._cache_size:
            self._cache, self._cache_size = {}, key
        return self._cache

    @property
    def state(self):
        host = np.array(self._h_state, dtype=np.float64).reshape(len(self._h_state), self.nstates)
        if self._src() is None:
            return host
        c = self._fresh()
        if "state" not in c:
            c["state"] = self._src().states()
        return np.vstack((c["state"], host)) if len(host) else c["state"]

    @property
    def pID(self):
        if self._src() is None:
            return self._h_pID
        c = self._fresh()
        if "pID" not in c:
            c["pID"] = self._src().parents().tolist()
        return c["pID"] + self._h_pID if self._h_pID else c["pID"]

    def _lqr_of(self, i):
        n = self._ndev()
        if i >= n:
            return self._h_lqr[i - n]
        c = self._fresh()
        if "K" not in c:
            c["K"] = self._src().gains()
        return (self._S, c["K"][i])

    def _edge(self, i):
        n = self._ndev()
        if i >= n:
            return (self._h_x[i - n], self._h_u[i - n])
        c = self._fresh()
        key = ("edge", i)
        if key not in c:
            x, u = self._src().edge(i)
            c[key] = (list(x), list(u))
        return c[key]

    # -- the reference's methods ----------------------------------------------------------------------
    def add_node(self, pID, state, lqr, x_seq, u_seq):
        """
        Adds a node to the tree (tree.py:77-96).  While the tree is growing on the device the waves append there
        (csrc k_append); this host-side append serves a tree built by hand and what the reference adds after
        planning (the finish_on_goal node, planner.py:299).
        """
        if pID >= self.size or pID < 0:
            raise ValueError("The given parent ID, {}, doesn't exist.".format(pID))
        self._h_state.append(np.array(state, dtype=np.float64))
        self._h_pID.append(int(pID))
        self._h_lqr.append(lqr)
        self._h_x.append(x_seq)
        self._h_u.append(u_seq)

    def climb(self, ID):
        """Node IDs from the seed (first element, 0) down to ID (last element) -- tree.py:100-117."""
        if ID >= self.size or ID < 0:
            raise ValueError("The given ID, {}, doesn't exist.".format(ID))
        if self._e is not None and "pID" not in self._fresh() and ID < self._ndev():
            # a tree that is still growing on the device: the engine's host mirror of the parents answers (the whole parent array of a
            # 200k-node tree is a 5 ms copy -- on the path of kill_update and at the end of every plan, tests/test_control_gpu.py)
            return self._dev().climb(ID)
        parents = self.pID
        chain = [int(ID)]
        while parents[chain[-1]] != -1:
            chain.append(int(parents[chain[-1]]))
        chain.reverse()
        return chain

    def trajectory(self, IDs):
        """(x_seq_full, u_seq_full): the edges of the listed nodes laid end to end -- tree.py:121-132."""
        xs, us = [], []
        src = self._src()
        if self._e is not None and len(IDs) > 2:
            # the whole path's edges in one gather (two copies) instead of two blocking copies per node
            c, n = self._fresh(), self._ndev()
            missing = [int(i) for i in IDs if int(i) < n and ("edge", int(i)) not in c]
            if missing:
                for i, (x, u) in zip(missing, src.edges_of(missing)):
                    c[("edge", i)] = (list(x), list(u))
        for ID in IDs:
            ex, eu = self._edge(int(ID))
            xs += list(ex)
            us += list(eu)
        return (xs, us)

    def visualize(self, dx, dy, node_seq=None, show=True):
        """
        Plots the (dx, dy) cross-section of the tree and highlights the path `node_seq` (tree.py:136-171).  Every edge
        is drawn from its parent's state through its recorded states; the edges come out of HBM in one bulk copy.
        Returns the matplotlib figure (show=False leaves it open for the caller).
        """
        from matplotlib import pyplot as plt
        from matplotlib.collections import LineCollection
        state, parents = self.state, self.pID
        on_path = set(int(i) for i in (node_seq if node_seq is not None else []))
        plain, marked = [], []
        for i in range(1, self.size):
            pts = np.vstack([state[parents[i]][None, :], np.array(self.x_seq[i], dtype=np.float64).reshape(-1, self.nstates)])
            (marked if i in on_path else plain).append(pts[:, [dx, dy]])
        fig, ax = plt.subplots()
        fig.suptitle("Tree")
        ax.add_collection(LineCollection(plain, colors="0.75", zorder=1))
        ax.add_collection(LineCollection(marked, colors="r", zorder=2))
        ax.scatter(state[0, dx], state[0, dy], color="b", s=48)
        if on_path:
            last = int(list(node_seq)[-1])
            ax.scatter(state[last, dx], state[last, dy], color="r", s=48)
        ax.set_xlabel("- State {} +".format(dx))
        ax.set_ylabel("- State {} +".format(dy))
        ax.grid(True)
        ax.autoscale()
        ax.set_aspect("equal", adjustable="datalim")
        if show:
            plt.show()
        return fig
