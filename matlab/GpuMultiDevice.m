% GpuMultiDevice.m
% Shim for libocs.so (include/ocs.h, ocs_multi_*): the batch axis of a call over the GPUs of one node.  The reference has
% no batch axis (tests/solve_test_problem.m:37 integrates one trajectory per call); here x0 is nStates x batch, u is
% nControls x (2N+1) x batch, and a call returns the arrays of the whole batch plus the RCCL-reduced statistics.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_multi_device.py).  See INTEGRATION.md.
classdef GpuMultiDevice < handle
   properties
      h          % libpointer to the ocs_multi handle
      devices    % HIP device ids, block k of a batch goes to devices(k)
   end
   methods
      function obj = GpuMultiDevice(devices)
         obj.devices = int32(devices(:)');  obj.h = libpointer('voidPtr');
         ocs_check(calllib('libocs', 'ocs_multi_create', obj.h, obj.devices, numel(devices)));
      end
      function delete(obj)
         calllib('libocs', 'ocs_multi_destroy', obj.h.Value);
      end
      function c = replicate(obj, make)
         % one handle object per device: make() runs with that device current (ocs_set_device)
         c = cell(1, numel(obj.devices));
         for k = 1:numel(obj.devices)
            ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(k)));  c{k} = make();
         end
         ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(1)));
      end
      function [x, J, stats] = compute_states(obj, integs, probs, x0, u)      % RK4Integrator.m:28-56 per trajectory
         batch = size(u, 3);  nAug = size(x0, 1) + 1;  N = (size(u, 2) - 1) / 2;
         x = zeros(nAug, N + 1, batch);  J = zeros(batch, 1);  stats = zeros(4, 1);
         gh = cellfun(@(g) g.hnd.Value, integs, 'UniformOutput', false);  ph = cellfun(@(p) p.h.Value, probs, 'UniformOutput', false);
         [rc, ~, ~, ~, ~, x, J, stats] = calllib('libocs', 'ocs_multi_compute_states', obj.h.Value, [gh{:}], [ph{:}], ...
               batch, x0, u, x, J, stats);
         ocs_check(rc);
      end
      % ---- device-resident blocks (ocs_multi_*_dev): the iterates of single_shooting.m:114,137-150 stay on the GPUs --------
      function blk = to_devices(obj, A, batch)
         % A: (rows x batch) or (rows x cols x batch) host array, trajectory index last (MATLAB shape).  Block k goes to
         % device k in the batch-minor layout of the _dev entry points; blk(k) = struct('p', device pointer, 'n', block size)
         per = numel(A) / batch;  A = reshape(A, per, batch);
         for k = 1:numel(obj.devices)
            lo = libpointer('int32Ptr', 0);  hi = libpointer('int32Ptr', 0);
            ocs_check(calllib('libocs', 'ocs_multi_shard', obj.h.Value, batch, k - 1, lo, hi));
            n = double(hi.Value - lo.Value);  bytes = uint64(8 * per * n);
            ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(k)));
            st = libpointer('voidPtr');  p = libpointer('voidPtr');
            ocs_check(calllib('libocs', 'ocs_device_malloc', st, bytes));  ocs_check(calllib('libocs', 'ocs_device_malloc', p, bytes));
            ocs_check(calllib('libocs', 'ocs_device_upload', st.Value, A(:, lo.Value + 1 : hi.Value), bytes, []));
            ocs_check(calllib('libocs', 'ocs_to_batch_minor_dev', st.Value, p.Value, per, n, []));
            ocs_check(calllib('libocs', 'ocs_synchronize'));  calllib('libocs', 'ocs_device_free', st.Value);
            blk(k) = struct('p', p.Value, 'n', n);  %#ok<AGROW>
         end
         ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(1)));
      end
      function blk = alloc_on_devices(obj, per, batch)
         blk = obj.to_devices(zeros(per, batch), batch);
      end
      function stats = nlp_objective_dev(obj, integs, probs, ctrls, x0blk, vblk, Jblk, dJdvblk)
         % [J, dJdv] = nlpObjective(v) on resident blocks; asynchronous, the RCCL reductions enqueued behind the kernels;
         % stats = {sum J, count, min J, index of the best candidate} (ocs_multi_stats waits for them)
         gh = cellfun(@(g) g.hnd.Value, integs, 'UniformOutput', false);  ph = cellfun(@(p) p.h.Value, probs, 'UniformOutput', false);
         ch = cellfun(@(c) c.h.Value, ctrls, 'UniformOutput', false);
         ocs_check(calllib('libocs', 'ocs_multi_nlp_objective_dev', obj.h.Value, [gh{:}], [ph{:}], [ch{:}], int32([vblk.n]), ...
               [x0blk.p], [vblk.p], 0, [], [Jblk.p], [dJdvblk.p], 1));
         stats = zeros(4, 1);
         [rc, ~, stats] = calllib('libocs', 'ocs_multi_stats', obj.h.Value, stats);  ocs_check(rc);
      end
      function A = from_devices(obj, blk, per)
         % the blocks back as one (per x batch) host array
         A = zeros(per, sum([blk.n]));  at = 0;
         for k = 1:numel(obj.devices)
            ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(k)));
            n = blk(k).n;  bytes = uint64(8 * per * n);  st = libpointer('voidPtr');
            ocs_check(calllib('libocs', 'ocs_device_malloc', st, bytes));
            ocs_check(calllib('libocs', 'ocs_to_traj_major_dev', blk(k).p, st.Value, per, n, []));
            part = zeros(per, n);
            [rc, part] = calllib('libocs', 'ocs_device_download', part, st.Value, bytes, []);  ocs_check(rc);
            calllib('libocs', 'ocs_device_free', st.Value);  A(:, at + 1 : at + n) = part;  at = at + n;
         end
         ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(1)));
      end
      function [J, dJdv, stats] = nlp_objective(obj, integs, probs, ctrls, x0, v)   % single_shooting.m:137-150
         batch = size(v, 2);  J = zeros(batch, 1);  dJdv = zeros(size(v));  stats = zeros(4, 1);
         gh = cellfun(@(g) g.hnd.Value, integs, 'UniformOutput', false);  ph = cellfun(@(p) p.h.Value, probs, 'UniformOutput', false);
         ch = cellfun(@(c) c.h.Value, ctrls, 'UniformOutput', false);
         [rc, ~, ~, ~, ~, ~, ~, J, dJdv, stats] = calllib('libocs', 'ocs_multi_nlp_objective', obj.h.Value, [gh{:}], [ph{:}], ...
               [ch{:}], batch, x0, v, 0, [], J, dJdv, stats);
         ocs_check(rc);
      end
   end
end
