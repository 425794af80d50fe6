% gpu_compute_equilibrium.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's functions/compute_equilibrium.m.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
function [xStar, lamStar, uStar, resnorm, residual, exitflag] = ...
   gpu_compute_equilibrium(prob, xGuess, lamGuess, uGuess, lb, ub, r)          % compute_equilibrium.m:1-2
   % xGuess, lamGuess: nStates x batch, uGuess: nControls x batch (batch = 1: the reference's call)
   nS = size(xGuess, 1);  nC = size(uGuess, 1);  batch = size(xGuess, 2);  n = 2*nS + nC;
   yG = [xGuess; lamGuess; uGuess];  y = zeros(n, batch);  residual = zeros(n, batch);
   resnorm = zeros(batch, 1);  exitflag = zeros(batch, 1, 'int32');
   [rc, ~, ~, ~, ~, y, resnorm, residual, exitflag] = calllib('libocs', 'ocs_compute_equilibrium', ...
         prob.h.Value, batch, r, yG, lb, ub, y, resnorm, residual, exitflag);
   ocs_check(rc);     % rc > 0 (OCS_NUM_NONFINITE): some instance has exitflag -1 (undefined values; lsqnonlin errors there)
   xStar = y(1:nS, :);  lamStar = y(nS+1:2*nS, :);  uStar = y(2*nS+1:end, :);   % :29-31
end
