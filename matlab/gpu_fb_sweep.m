% gpu_fb_sweep.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's functions/fb_sweep.m.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
function soln = gpu_fb_sweep(prob, x0, tspan, options)                         % fb_sweep.m:1
   integ = GpuRK4Integrator(tspan);  o = libstruct('ocs_fbs_options');
   calllib('libocs', 'ocs_fbs_default_options', o);                           % :16-22
   names = {'uRelTol', 'uAbsTol', 'nSWEEPS', 'nERROR_PTS', 'nINTERP_PTS', ...          % :26-59
            'uRelax'};       % extension: damped update u = u + uRelax (uNew - u); 0 = the reference's u = uNew (:85)
   if nargin > 3
      for k = 1:numel(names)
         if isfield(options, names{k}), o.(names{k}) = options.(names{k}); end
      end
   end
   % (the other fields are build options and keep their defaults: fused_update_off, nWINDOWS, cost_row -- the last
   %  only matters to callers of ocs_fb_sweep_dev that want the running objective at every node)
   N = numel(tspan) - 1;  nS = numel(x0);  nC = size(prob.ControlBounds, 1);
   x = zeros(nS, N+1);  lam = x;  uI = zeros(nC, o.nINTERP_PTS);  J = 0;  sweeps = int32(0);
   [rc, ~, ~, ~, ~, ~, ~, x, lam, uI, J, sweeps] = calllib('libocs', 'ocs_fb_sweep', integ.hnd.Value, ...
         prob.h.Value, 1, x0, o, [], [], x, lam, uI, J, sweeps, []);
   if rc < 0, ocs_check(rc); end   % bad options, unsupported problem, HIP error; rc > 0 = OCS_NUM_NOT_CONVERGED:
   soln = struct();                                                            % stays empty (:77)
   if sweeps > 0
      interpPts = linspace(tspan(1), tspan(end), o.nINTERP_PTS);
      soln.x = vectorInterpolant(tspan, x, 'pchip');  soln.lam = vectorInterpolant(tspan, lam, 'pchip');
      soln.u = vectorInterpolant(interpPts, uI, 'pchip');  soln.J = J;         % :82, :123
   end
end
