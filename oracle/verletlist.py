"""ORACLE — TEST INFRASTRUCTURE ONLY.

Host logic of the reference's Verlet list restated in Python on top of the C kernels of oracle/src/lj.c:
  BasicNeighbourListBase::{update, fillBasicNeighbourList, tryToFillNeighbourList, increaseMaximumNeighboursPerParticle}
                                  Interactor/NeighbourList/BasicList/BasicListBase.cuh:76-215   (capacity starts at 32, +32)
  VerletListBase::{update, needsRebuild, isParticleDriftOverThreshold, storeCurrentPos, updateSortedPositions, rebuildList}
                                  Interactor/NeighbourList/VerletList/VerletListBase.cuh:73-199 (multiplier 1.08,
                                  threshold (1.08 rc - rc)/2, `thresholdDistance <= 1e-6` -> always rebuild)
  VerletList::{update, needsRebuild, handlePosWriteRequested, handleReorder}
                                  Interactor/NeighbourList/VerletList.cuh:112-124, :171-200
"""
import numpy as np


class VerletListOracle:
    def __init__(self, oracle):
        self.o = oracle
        self.real = oracle.real
        # BasicNeighbourListBase
        self.maxNeighboursPerParticle = 32
        # VerletListBase
        self.verletRadiusMultiplier = self.real(1.08)
        self.currentCutOff = self.real(0)
        self.currentBox = None
        self.storedPos = None
        self.forceNextRebuild = True
        self.stepsSinceLastUpdate = 0
        # VerletList
        self.forceNextUpdate = True
        self.wrapCutOff, self.wrapBox = None, None
        self.rebuilds = 0

    # -- VerletList ------------------------------------------------------------------------------------
    def handlePosWriteRequested(self):
        self.forceNextUpdate = True

    def handleReorder(self):
        self.forceNextUpdate = True
        self.forceNextRebuild = True

    def setCutOffMultiplier(self, m):
        self.forceNextUpdate = True
        self.forceNextRebuild = True
        self.verletRadiusMultiplier = self.real(m)

    def getNumberOfStepsSinceLastUpdate(self):
        return self.stepsSinceLastUpdate - 1

    def update(self, pos, box_L, box_periodic, cutOff):
        """VerletList::update: only reaches the base when positions were written / box or cut-off changed."""
        box = (tuple(np.broadcast_to(np.asarray(box_L, dtype=self.real), (3,)).tolist()),
               tuple(int(x) for x in np.broadcast_to(np.asarray(box_periodic), (3,))))
        cutOff = self.real(cutOff)
        need = self.forceNextUpdate or box != self.wrapBox or cutOff != self.wrapCutOff
        self.forceNextUpdate = False
        if need:
            self.wrapBox, self.wrapCutOff = box, cutOff
            self._base_update(self.o.r(pos), box, cutOff)

    # -- VerletListBase --------------------------------------------------------------------------------
    def _needs_rebuild(self, pos, box, cutOff):
        if self.forceNextRebuild:
            self.forceNextRebuild = False
            return True
        if box != self.currentBox or cutOff != self.currentCutOff or len(pos) != len(self.storedPos):
            return True
        # isParticleDriftOverThreshold: real arithmetic, `/ 2.0` promotes to double in the reference
        threshold = self.real((float(self.verletRadiusMultiplier * self.currentCutOff) - float(self.currentCutOff)) / 2.0) \
            if self.real == np.float64 else self.real((np.float32(self.verletRadiusMultiplier * self.currentCutOff) -
                                                       self.currentCutOff).astype(np.float64) / 2.0)
        if threshold <= 1e-6:
            return True
        return self.o.verletlist_check_drift(pos, self.storedPos, threshold, box[0], box[1]) > 0

    def _base_update(self, pos, box, cutOff):
        if self._needs_rebuild(pos, box, cutOff):
            self.stepsSinceLastUpdate = 0
            self.currentBox, self.currentCutOff = box, cutOff
            self.storedPos = pos.copy()
            self._rebuild()
        self.sortPos = np.ascontiguousarray(pos[self.groupIndex])        # updateSortedPositions
        self.stepsSinceLastUpdate += 1

    # -- BasicNeighbourListBase ------------------------------------------------------------------------
    def _rebuild(self):
        self.rebuilds += 1
        o = self.o
        rcut = self.real(self.currentCutOff * self.verletRadiusMultiplier)
        L, per = self.currentBox
        cd, gL, gper = o.celllist_create_grid(L, per, rcut)
        cl = o.celllist_build(self.storedPos, gL, gper, cd)
        assert cl["error"] == 0
        n = len(self.storedPos)
        while True:
            flag, nl, nn = o.verletlist_fill(cl, L, per, self.real(rcut * rcut), self.maxNeighboursPerParticle, n)
            if flag == 0:
                break
            self.maxNeighboursPerParticle += 32
        self.cl, self.neighbourList, self.numberNeighbours, self.groupIndex = cl, nl, nn, cl["index"]

    # -- traversal -------------------------------------------------------------------------------------
    def lj_forces(self, box_L, box_periodic, param_table, ntypes, **kw):
        return self.o.lj_transverse_verletlist(self.sortPos, self.groupIndex, self.neighbourList, self.numberNeighbours,
                                               box_L, box_periodic, param_table, ntypes, **kw)
